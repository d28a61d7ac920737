#!/bin/bash
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?}"
TAG=${1:-r02_bench}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "host_memory or saturate" > "$OUT/pytest.log" 2>&1 || { tail -40 "$OUT/pytest.log"; exit 1; }
tail -2 "$OUT/pytest.log"
nproc; free -g | head -2
timeout -k 10 600 python bench.py --steps 20 --warmup 3 > "$OUT/bench_line.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
python - <<'PY'
import json,sys
d=json.loads(open(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r02_bench/bench_line.json").read().strip().split("\n")[-1])
print({k:d[k] for k in ("value","ms_per_step","value_host_pinned","ms_per_step_host_pinned")})
print(d["cpu_baseline"]); print({k:v for k,v in d["roofline"].items() if k!="note"})
PY
