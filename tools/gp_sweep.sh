#!/bin/bash
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/gp
for R in 1024 512 256 128; do
  SHK_GP_ROWS=$R python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-leg 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); s=d['stage_ms']; print('gp rows target $R', round(d['ms_per_step'],3), {k: round(s[k],3) for k in ('graph_table_kernel','adjacency_kernel','assemble_device_total_host_clock')})"
done | tee gpurun_out/gp/sweep.txt
