#!/bin/bash
# round 4, pass 1: count step timing (shipped build), then the in-kernel clocks and the ablations (ABLATE build)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
for win in ${1:-18}; do echo "== SHK_PART_WIN=$win"; SHK_PART_WIN=$win timeout -k 10 200 python tools/pre_only.py 2>&1 | tail -1; done
cp sparrowhawk_amd/libshk_hip_ablate.so sparrowhawk_amd/libshk_hip.so      # (on the GPU box's copy of the tree only)
for D in 0 2; do
  echo "== ABLATE SHK_DEBUG_NOSTORE=$D"; SHK_DEBUG_P1CLK=1 SHK_DEBUG_NOSTORE=$D timeout -k 10 120 python3 tools/pre_only.py 2>&1 | grep -E "p1clk|partition_kernel" | tail -2
done
echo "== ABLATE no clocks"; for D in 0 1 2; do SHK_DEBUG_NOSTORE=$D timeout -k 10 120 python3 tools/pre_only.py 2>&1 | grep -E "partition_kernel" | tail -1; done
