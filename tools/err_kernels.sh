#!/bin/bash
# per-kernel times of the error-rich workload (errors left in)
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=$PWD/gpurun_out/errk; mkdir -p "$OUT"; export TMPDIR=/tmp
ARGS=${*:-"--err 0.005"}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o s -- python bench.py $ARGS --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg > "$OUT/log.txt" 2>&1
find "$OUT" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*.db" -delete; find "$OUT" -name "*agent_info.csv" -delete
python - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "at::native" not in r["Name"]]
for r in rows[:16]:
    print("%-62s calls %5s avg %9.1f us" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
