# error-rich configs[2] (k=51, 1 % errors left in): k_count_buckets timing experiments (SHK_DEBUG_B)
set -e
mkdir -p gpurun_out/cfg3
for B in 0 1 2 3; do
  ERR=0.01 K=51 SHK_DEBUG_B=$B timeout -k 10 300 python tools/pre_only.py > gpurun_out/cfg3/b$B.log 2>&1 || true
  echo "SHK_DEBUG_B=$B: $(tail -1 gpurun_out/cfg3/b$B.log)"
done
