#!/bin/bash
# round 4 quick GPU check: parity tests (optional), a bench line, the idle gaps of a step (one handle at a time).
# Usage (via gpurun): bash tools/r04_quick.sh <tag> [tests|notests|fast] [bench args...]
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
TAG=${1:-r04q}; TESTS=${2:-tests}; shift; shift || true
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
if [ "$TESTS" = tests ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=8 > "$OUT/pytest.log" 2>&1 || { tail -60 "$OUT/pytest.log"; exit 1; }
  tail -14 "$OUT/pytest.log"
elif [ "$TESTS" = fast ]; then
  timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --durations=5 > "$OUT/pytest.log" 2>&1 || { tail -60 "$OUT/pytest.log"; exit 1; }
  tail -8 "$OUT/pytest.log"
fi
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > "$OUT/bench_line.json" 2> "$OUT/bench.err" || { tail -30 "$OUT/bench.err"; exit 1; }
python tools/bench_summary.py "$OUT/bench_line.json" || true
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace1" -o t -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --no-legs --no-inflight-leg > "$OUT/trace1.log" 2>&1 || true
find "$OUT/trace1" -name "*kernel_trace.csv" -exec python tools/gap_report.py {} \; > "$OUT/gpu_idle_gaps.txt" 2>&1 || true
head -24 "$OUT/gpu_idle_gaps.txt"
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete; find "$OUT" -name "*.db" -delete
