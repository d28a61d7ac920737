# two-kernel pass 2 (k_dedupe_partitions + k_count_weighted): timing experiments on the bench workload (ABLATE build)
set -e
mkdir -p gpurun_out/split
cp sparrowhawk_amd/libshk_hip_ablate.so sparrowhawk_amd/libshk_hip.so      # (on the GPU box's copy of the tree only)
for D in 0 21 22 23 31 32 33; do
  echo "SHK_DEBUG_P2=$D: $(ERR=0 K=31 SHK_DEBUG_P2=$D timeout -k 10 120 python3 tools/pre_only.py 2>&1 | tail -n 1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print({k: d[k] for k in d if k in ('count_kernel','count_dedupe_kernel')})")"
done 2>&1 | tee gpurun_out/split/ablate.txt
