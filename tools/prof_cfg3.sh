#!/bin/bash
set -euo pipefail
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/cfg3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cfg3/prof -o s -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --k 51 --err 0.01 > gpurun_out/cfg3/prof.log 2>&1
find gpurun_out/cfg3/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/cfg3/kernel_stats_unmasked.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cfg3/prof2 -o s -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --k 51 --err 0.01 --mask-errors > gpurun_out/cfg3/prof2.log 2>&1
find gpurun_out/cfg3/prof2 -name "*kernel_stats.csv" -exec cp {} gpurun_out/cfg3/kernel_stats_masked.csv \;
rm -rf gpurun_out/cfg3/prof gpurun_out/cfg3/prof2
