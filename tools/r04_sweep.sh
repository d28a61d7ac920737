#!/bin/bash
# round 4: re-tune the assembly's knobs behind the new pass 1 / one partition per table (whole step, quiet handle)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
run() { echo "$1 $(env $1 timeout -k 10 200 python tools/step_breakdown.py 2>&1 | grep -E "^(preprocess|assemble|sum)" | tr '\n' ' ')"; }
run "SHK_GP_ROWS=640"; run "SHK_GP_ROWS=320"; run "SHK_GP_ROWS=1280"; run "SHK_GP_ROWS=480"
run "SHK_TILE_ROWS=2560"; run "SHK_TILE_ROWS=1280"; run "SHK_TILE_ROWS=5120"
run "SHK_SPLIT_LOG=6"; run "SHK_SPLIT_LOG=5"; run "SHK_SPLIT_LOG=7"
run "SHK_GP_ROWS=640"
