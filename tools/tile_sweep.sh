#!/bin/bash
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/tile
for R in 1536 2048 2560 3072 3584 4096; do
  SHK_TILE_ROWS=$R python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-leg 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); s=d['stage_ms']; print('tile rows $R', round(d['ms_per_step'],3), {k: round(s[k],3) for k in ('collapse_succ_split','collapse_walk','collapse_rank_device','collapse_emit','assemble_device_total_host_clock')})"
done | tee gpurun_out/tile/sweep.txt
