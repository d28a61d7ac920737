#!/bin/bash
# FASTQ entry point: number of pieces of a single-batch text
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/fq
for C in 1 2 4 8 16; do
  echo "== SHK_FASTQ_PIECES=$C"
  SHK_FASTQ_PIECES=$C SKIP_HOST=1 ONLY_DEVICE=1 timeout -k 10 300 python tools/fastq_path.py 2>&1 | grep "run 1"
done | tee gpurun_out/fq/pieces.txt
