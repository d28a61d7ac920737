#!/bin/bash
# round 4: wave-cycles per phase of k_partition (ABLATE build, SHK_DEBUG_P1CLK=1)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
cp sparrowhawk_amd/libshk_hip_ablate.so sparrowhawk_amd/libshk_hip.so      # (on the GPU box's copy of the tree only)
for D in 0 2; do
  echo "== SHK_DEBUG_NOSTORE=$D"; SHK_DEBUG_P1CLK=1 SHK_PART_WIN=${1:-18} SHK_DEBUG_NOSTORE=$D timeout -k 10 120 python3 tools/pre_only.py 2>&1 | grep -E "p1clk|partition_kernel" | tail -3
done
