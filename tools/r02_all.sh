#!/bin/bash
# whole GPU suite + bench line (hard timeouts: a hang must not eat the box)
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?}"
TAG=${1:-r02_all}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=12 > "$OUT/pytest.log" 2>&1 || { tail -60 "$OUT/pytest.log"; exit 1; }
tail -16 "$OUT/pytest.log"
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_line.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
python tools/bench_summary.py "$OUT/bench_line.json"
