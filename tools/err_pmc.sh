#!/bin/bash
# HBM traffic of the bucket path (error-rich reads): FETCH_SIZE and WRITE_SIZE in separate passes
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=$PWD/gpurun_out/errpmc; mkdir -p "$OUT"; export TMPDIR=/tmp
ARGS=${*:-"--err 0.005"}
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o p -- python bench.py $ARGS --steps 2 --warmup 1 --no-cpu-baseline --no-host-leg > "$OUT/f.log" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write" -o p -- python bench.py $ARGS --steps 2 --warmup 1 --no-cpu-baseline --no-host-leg > "$OUT/w.log" 2>&1
python tools/pmc_summary.py "$OUT/pmc_*/**/*counter_collection.csv" > "$OUT/pmc_summary.txt" 2>&1 || true
find "$OUT" -name "*counter_collection.csv" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*.db" -delete; find "$OUT" -name "*agent_info.csv" -delete
grep -E "k_ovf_scatter|k_count_buckets|k_count_partitions|k_partition" "$OUT/pmc_summary.txt" | head -20
