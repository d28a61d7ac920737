#!/bin/bash
# round 4: fast parity set + stage times of a step (stage timers on), then the quiet step
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
TAG=${1:-r04g}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
if [ "${2:-fast}" = fast ]; then
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > "$OUT/pytest.log" 2>&1 || { tail -60 "$OUT/pytest.log"; exit 1; }
tail -2 "$OUT/pytest.log"
fi
SHK_STAGE_TIMERS=1 timeout -k 10 200 python tools/step_breakdown.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/stages.txt" | grep -E "sum|partition_kernel|count_kernel|graph|adj|collapse|correct" 
for i in 1 2; do timeout -k 10 200 python tools/step_breakdown.py 2>&1 | grep -E "^(preprocess|assemble|get_assembly|sum)" | tr '\n' ' '; echo; done
