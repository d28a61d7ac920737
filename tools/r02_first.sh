#!/bin/bash
# round 2, first GPU call: parity tests at HEAD, the bench line, and the N=2 rehearsals of the sharded path
# (RCCL inside the library with both ranks on the one GPU — expected to be refused as a duplicate GPU — and
# the torch/gloo-driven variant).
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?}"
OUT=$PWD/gpurun_out/r02_first; mkdir -p "$OUT"
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=10 > "$OUT/pytest.log" 2>&1 || { tail -60 "$OUT/pytest.log"; exit 1; }
tail -16 "$OUT/pytest.log"
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_line.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
python tools/bench_summary.py "$OUT/bench_line.json"
echo "--- N=2 one GPU, RCCL inside the library"
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29811 bench.py --gpus 2 --steps 3 --warmup 1 --one-gpu --backend gloo > "$OUT/n2_lib.json" 2> "$OUT/n2_lib.err" || { echo "rc=$?"; tail -5 "$OUT/n2_lib.err"; }
echo "--- N=2 one GPU, torch/gloo-driven collectives"
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29812 bench.py --gpus 2 --steps 3 --warmup 1 --one-gpu --backend gloo --collectives torch > "$OUT/n2_torch.json" 2> "$OUT/n2_torch.err" || { echo "rc=$?"; tail -5 "$OUT/n2_torch.err"; }
python tools/bench_summary.py "$OUT/n2_torch.json" || true
