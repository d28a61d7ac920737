#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
TAG=${1:-r04g2}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_dist.py -m gpu -x -q > "$OUT/pytest.log" 2>&1 || { tail -60 "$OUT/pytest.log"; exit 1; }
tail -2 "$OUT/pytest.log"
bash tools/r04_stats.sh $TAG | head -20
for i in 1 2; do timeout -k 10 200 python tools/step_breakdown.py 2>&1 | grep -E "^(preprocess|assemble|get_assembly|sum)" | tr '\n' ' '; echo; done
