#!/bin/bash
# round 4: the sharded path with a one-rank communicator under the kernel trace: where the GPU idles inside a step
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=$PWD/gpurun_out/${1:-r04shgap}; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o t -- python bench.py ${BENCH_ARGS:---force-sharded} --steps 6 --warmup 2 --no-cpu-baseline --no-host-leg --no-legs --no-inflight-leg > "$OUT/trace.log" 2>&1 || tail -5 "$OUT/trace.log"
F=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
starts = [i for i, e in enumerate(ev) if "k_partition" in e[2]]
a, b = starts[-2], starts[-1]
step = ev[a:b]; t0 = step[0][0]
print("step span %.3f ms, busy %.3f ms, %d kernels" % ((ev[b][0] - t0) / 1e6, sum(e[1] - e[0] for e in step) / 1e6, len(step)))
for i, e in enumerate(step):
    nxt = step[i + 1][0] if i + 1 < len(step) else ev[b][0]
    print("%8.1f us  +%7.1f run %7.1f gap  %s" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, (nxt - e[1]) / 1e3, e[2].split("(")[0][-48:]))
PY
find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*agent_info.csv" -delete; find "$OUT" -name "*.db" -delete
