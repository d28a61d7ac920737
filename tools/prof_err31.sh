#!/bin/bash
set -euo pipefail
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/err31
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/err31/prof -o s -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --err 0.005 > gpurun_out/err31/prof.log 2>&1
find gpurun_out/err31/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/err31/kernel_stats.csv \;
rm -rf gpurun_out/err31/prof
