# count-step timings of the error-rich configurations (tools/pre_only.py)
set -e
echo "k=51 1%: $(ERR=0.01 K=51 timeout -k 10 300 python tools/pre_only.py 2>&1 | tail -1)"
echo "k=31 0.5%: $(ERR=0.005 K=31 timeout -k 10 300 python tools/pre_only.py 2>&1 | tail -1)"
echo "k=31 0.1%: $(ERR=0.001 K=31 timeout -k 10 300 python tools/pre_only.py 2>&1 | tail -1)"
echo "k=31 clean: $(timeout -k 10 300 python tools/pre_only.py 2>&1 | tail -1)"
