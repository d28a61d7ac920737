#!/bin/bash
set -euo pipefail
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/gap
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap/prof -o s -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-host-leg > gpurun_out/gap/prof.log 2>&1
python tools/gap_report.py $(find gpurun_out/gap/prof -name "*kernel_trace.csv" | head -1) | tee gpurun_out/gap/gpu_idle_gaps.txt
rm -rf gpurun_out/gap/prof
