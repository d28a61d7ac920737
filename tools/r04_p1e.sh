#!/bin/bash
# round 4, pass 1: one named test first, then the fast parity set, then the count step timing
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
TAG=${1:-r04p1e}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mixed_lengths or long_reads" > "$OUT/pytest_first.log" 2>&1 || { tail -40 "$OUT/pytest_first.log"; exit 1; }
tail -2 "$OUT/pytest_first.log"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q --durations=5 > "$OUT/pytest.log" 2>&1 || { tail -60 "$OUT/pytest.log"; exit 1; }
tail -3 "$OUT/pytest.log"
for i in 1 2; do timeout -k 10 200 python tools/pre_only.py 2>&1 | tail -1 | cut -c1-120; done
