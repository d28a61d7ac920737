#!/bin/bash
# round 4: per-kernel times of the bench step (rocprofv3 --kernel-trace --stats), top of the list
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
TAG=${1:-r04stats}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -o s -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-leg --no-legs --no-inflight-leg > "$OUT/prof_stats.log" 2>&1
find "$OUT/prof_stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = [int(r['Calls']) for r in rows if 'k_partition' in r['Name']][0]
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:34]:
    if 'at::native' in r['Name']: continue
    print("%-60s calls/step %5.1f  per-step %8.1f us" % (r['Name'][:60], int(r['Calls']) / steps, float(r['TotalDurationNs']) / steps / 1e3))
PY
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete; find "$OUT" -name "*.db" -delete
