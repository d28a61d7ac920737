#!/bin/bash
# rocprofv3 kernel stats of the FASTQ-text entry point (device parser), 1.05 GB of text
set -euo pipefail
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/fastq
SKIP_HOST=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fastq/prof -o s -- python tools/fastq_path.py > gpurun_out/fastq/prof.log 2>&1
find gpurun_out/fastq/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/fastq/kernel_stats_fastq.csv \;
rm -rf gpurun_out/fastq/prof
tail -8 gpurun_out/fastq/prof.log
