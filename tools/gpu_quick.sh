#!/bin/bash
# quick GPU check: parity tests (optional) + one bench line.  Usage: bash tools/gpu_quick.sh <tag> [tests|notests] [bench args...]
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?}"
TAG=${1:-quick}; TESTS=${2:-tests}; shift; shift || true
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
if [ "$TESTS" = tests ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=8 > "$OUT/pytest.log" 2>&1 || { tail -40 "$OUT/pytest.log"; exit 1; }
  tail -14 "$OUT/pytest.log"
fi
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > "$OUT/bench_line.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
python tools/bench_summary.py "$OUT/bench_line.json"
