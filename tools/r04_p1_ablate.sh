#!/bin/bash
# round 4: where k_partition's time goes (ABLATE build: SHK_DEBUG_NOSTORE 1 = every record to slot 0 of its slice, 2 = no flush)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
cp sparrowhawk_amd/libshk_hip_ablate.so sparrowhawk_amd/libshk_hip.so      # (on the GPU box's copy of the tree only)
for win in ${1:-18}; do for D in 0 1 2; do
  echo "WIN=$win SHK_DEBUG_NOSTORE=$D: $(SHK_PART_WIN=$win SHK_DEBUG_NOSTORE=$D timeout -k 10 120 python3 tools/pre_only.py 2>&1 | tail -n 1 | python3 -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print({k: d[k] for k in d if k in ('partition_kernel',)})")"
done; done
