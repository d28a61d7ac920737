#!/bin/bash
# round 4: the minimiser window (SHK_PART_WIN) against every stage of a step (stage timers on), then the step without timers
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
TAG=${1:-r04win}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
for win in ${2:-16 18 20}; do
  echo "== SHK_PART_WIN=$win (stage timers)"; SHK_PART_WIN=$win SHK_STAGE_TIMERS=1 timeout -k 10 200 python tools/step_breakdown.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/stages_win$win.txt" | grep -E "sum|partition_kernel|count_kernel|graph|adj|collapse_(walk|frag|succ|rank|emit|final)|correct" 
  echo "== SHK_PART_WIN=$win (quiet)"; SHK_PART_WIN=$win timeout -k 10 200 python tools/step_breakdown.py 2>&1 | grep -E "^(preprocess|assemble|get_assembly|sum)"
done
