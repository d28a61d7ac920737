#!/bin/bash
# round 4, pass 1: what the parts of the flush cost (ABLATE build, SHK_DEBUG_NOSTORE 0 / 2 = no flush / 3 = no cursor atomic /
# 4 = no bases, no mask / 5 = no store / 6 = no mask row)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
cp sparrowhawk_amd/libshk_hip_ablate.so sparrowhawk_amd/libshk_hip.so      # (on the GPU box's copy of the tree only)
for D in ${1:-0 2 3 4 5 6}; do echo "SHK_DEBUG_NOSTORE=$D $(SHK_DEBUG_NOSTORE=$D timeout -k 10 120 python3 tools/pre_only.py 2>&1 | grep -E "partition_kernel" | tail -1 | cut -c1-60)"; done
