#!/bin/bash
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=$PWD/gpurun_out/r02_bloom; mkdir -p "$OUT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -s -k "bloom or exceed or progress" > "$OUT/pytest.log" 2>&1 || { tail -60 "$OUT/pytest.log"; exit 1; }
grep -E "configs\[2\]|passed|failed" "$OUT/pytest.log"
