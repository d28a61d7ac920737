#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
run() { echo "$1 $(env $1 timeout -k 10 200 python tools/step_breakdown.py 2>&1 | grep -E "^(assemble|sum)" | tr '\n' ' ')"; }
for rep in 1 2; do for T in 2560 640 1024 1280 1536 2048; do run "SHK_TILE_ROWS=$T"; done; done
