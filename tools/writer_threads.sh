#!/bin/bash
# host-side jitter of the step against the number of writer threads
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?}"
mkdir -p gpurun_out/wt
{
echo "nproc $(nproc)  cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null || echo n/a)  affinity $(taskset -p $$ 2>/dev/null | tail -c 40)"
for T in 16 12 8 6 4; do
  for r in 1 2 3; do
    SHK_WRITER_THREADS=$T timeout -k 10 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-host-leg 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); s=d['stage_ms']; print('threads $T run $r: %.3f ms/step  asm_dev %.3f outputs %.3f pre %.3f' % (d['ms_per_step'], s['assemble_device_total_host_clock'], s['outputs_host_clock'], s['preprocess_device_total_host_clock']))"
  done
done
} | tee gpurun_out/wt/threads.txt
