set -e
mkdir -p gpurun_out/split
for cfg in "ERR=0 K=31" "ERR=0.005 K=31" "ERR=0.001 K=31" "ERR=0.01 K=51" "ERR=0.01 K=51 MASK=1" "ERR=0 K=89"; do
  for S in 1 0; do
    echo "$cfg split=$S: $(env $cfg SHK_COUNT_SPLIT=$S timeout -k 10 120 python3 tools/pre_only.py 2>&1 | tail -n 1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print({k: d[k] for k in d if k in ('count_kernel','count_dedupe_kernel','count_repartitioned_x1','count_deferred_untried_x1','preprocess_device_total_host_clock')})")"
  done
done 2>&1 | tee gpurun_out/split/check.txt
