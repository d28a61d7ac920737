# two-kernel pass 2: partitions counted together per table (SHK_COUNT_MERGE), fused kernel beside it
set -e
mkdir -p gpurun_out/split
for M in 1 2 3 4; do
  echo "merge=$M: $(ERR=0 K=31 SHK_COUNT_MERGE=$M timeout -k 10 120 python3 tools/pre_only.py 2>&1 | tail -n 1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print({k: d[k] for k in d if k in ('count_kernel','count_dedupe_kernel','count_repartitioned_x1','preprocess_device_total_host_clock')})")"
done 2>&1 | tee gpurun_out/split/merge.txt
echo "fused: $(ERR=0 K=31 SHK_COUNT_SPLIT=0 timeout -k 10 120 python3 tools/pre_only.py 2>&1 | tail -n 1)" | tee -a gpurun_out/split/merge.txt
echo "k51 masked merge 2: $(ERR=0.01 MASK=1 K=51 timeout -k 10 120 python3 tools/pre_only.py 2>&1 | tail -n 1)" | tee -a gpurun_out/split/merge.txt
echo "k51 masked fused: $(ERR=0.01 MASK=1 K=51 SHK_COUNT_SPLIT=0 timeout -k 10 120 python3 tools/pre_only.py 2>&1 | tail -n 1)" | tee -a gpurun_out/split/merge.txt
echo "k31 err 0.5% merge 2: $(ERR=0.005 K=31 timeout -k 10 120 python3 tools/pre_only.py 2>&1 | tail -n 1)" | tee -a gpurun_out/split/merge.txt
echo "k31 err 0.5% fused: $(ERR=0.005 K=31 SHK_COUNT_SPLIT=0 timeout -k 10 120 python3 tools/pre_only.py 2>&1 | tail -n 1)" | tee -a gpurun_out/split/merge.txt
