#!/bin/bash
# randomised differential campaign on a GPU box.  Usage: bash tools/fuzz_round.sh <tag> <cases> <seed> [genome size range]
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?}"
TAG=${1:-fuzz}; N=${2:-300}; SEED=${3:-1}; LO=${4:-300}; HI=${5:-30000}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
timeout -k 10 900 python tools/fuzz_parity.py "$N" "$SEED" "$LO" "$HI" > "$OUT/fuzz.txt" 2>&1 || { tail -30 "$OUT/fuzz.txt"; exit 1; }
tail -2 "$OUT/fuzz.txt"
