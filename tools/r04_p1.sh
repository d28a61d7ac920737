#!/bin/bash
# round 4, pass-1 work: parity (fast set), then the count step alone under the window settings, then a bench line.
# Usage (via gpurun): bash tools/r04_p1.sh <tag> [fast|notests|tests]
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
TAG=${1:-r04p1}; TESTS=${2:-fast}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
if [ "$TESTS" = tests ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=8 > "$OUT/pytest.log" 2>&1 || { tail -60 "$OUT/pytest.log"; exit 1; }
  tail -5 "$OUT/pytest.log"
elif [ "$TESTS" = fast ]; then
  timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --durations=5 > "$OUT/pytest.log" 2>&1 || { tail -60 "$OUT/pytest.log"; exit 1; }
  tail -5 "$OUT/pytest.log"
fi
for win in 20 16; do
  echo "== SHK_PART_WIN=$win"; SHK_PART_WIN=$win timeout -k 10 200 python tools/pre_only.py 2>&1 | tail -2
done
echo "== K=51 masked"; K=51 ERR=0.01 MASK=1 timeout -k 10 200 python tools/pre_only.py 2>&1 | tail -1
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-legs > "$OUT/bench_line.json" 2> "$OUT/bench.err" || { tail -30 "$OUT/bench.err"; exit 1; }
python tools/bench_summary.py "$OUT/bench_line.json" || true
