#!/bin/bash
# collapse experiments: GPU tests (optional), bench line, per-kernel stats.  Usage: bash tools/frag_round.sh <tag> [tests|notests]
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?}"
TAG=${1:-frag}; TESTS=${2:-tests}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
if [ "$TESTS" = tests ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest.log" 2>&1 || { tail -40 "$OUT/pytest.log"; exit 1; }
  tail -3 "$OUT/pytest.log"
fi
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-leg > "$OUT/bench_line.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
python tools/bench_summary.py "$OUT/bench_line.json"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -o s -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-leg > "$OUT/prof_stats.log" 2>&1
find "$OUT" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete; find "$OUT" -name "*.db" -delete
python - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:32]:
    print("%-60s calls %5s avg %9.1f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
