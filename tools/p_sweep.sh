set -e
for P in ${PS:-4096 8192 16384}; do
for E in 0 0.001 0.003; do
  echo "P=$P err=$E: $(SHK_PART_P=$P ERR=$E K=31 timeout -k 10 300 python tools/pre_only.py 2>&1 | tail -1 | python -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print({k: d[k] for k in d if k in ('count_kernel','partition_kernel','count_repartitioned_x1','preprocess_device_total_host_clock')})")"
done; done
