#!/bin/bash
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=$PWD/gpurun_out/r02_sweep; mkdir -p "$OUT"
for sl in 4 5 6 7 8; do
  SHK_SPLIT_LOG=$sl timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/split_$sl.json" 2> "$OUT/split_$sl.err"
  echo -n "SPLIT_LOG=$sl  "; python tools/bench_summary.py "$OUT/split_$sl.json"
done
