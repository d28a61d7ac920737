#!/bin/bash
# round 4, pass 1: descriptors a lane may hold before its wave flushes (ABLATE build, SHK_DEBUG_P1FLUSH; 0 = the default 9 .. 6)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
cp sparrowhawk_amd/libshk_hip_ablate.so sparrowhawk_amd/libshk_hip.so      # (on the GPU box's copy of the tree only)
for F in ${1:-0 1 2 3 5 7 9 0}; do echo "SHK_DEBUG_P1FLUSH=$F $(SHK_DEBUG_P1FLUSH=$F timeout -k 10 120 python3 tools/pre_only.py 2>&1 | grep -E "partition_kernel" | tail -1 | cut -c1-60)"; done
