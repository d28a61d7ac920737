#!/bin/bash
# where the time goes on reads that carry errors (k=31, 5 Mbp, 100x): masked by quality, and left in
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/errp
for ARGS in "--err 0.005 --mask-errors" "--err 0.005" "--err 0.01"; do
  python bench.py $ARGS --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); s=d['stage_ms']; print('$ARGS', round(d['ms_per_step'],3), 'ms/step', {k: round(v,3) for k,v in sorted(s.items()) if v > 0.05 and 'x1' not in k})"
done | tee gpurun_out/errp/profile.txt
