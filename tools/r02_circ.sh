#!/bin/bash
# circular unitigs on the device: the parity cases and the full-size isolate + plasmid
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=$PWD/gpurun_out/r02_circ; mkdir -p "$OUT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -s --durations=8 > "$OUT/pytest.log" 2>&1 || { tail -60 "$OUT/pytest.log"; exit 1; }
grep -E "assemble \(device|passed|failed" "$OUT/pytest.log"
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_line.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
python tools/bench_summary.py "$OUT/bench_line.json"
