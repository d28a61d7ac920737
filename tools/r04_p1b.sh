#!/bin/bash
# round 4, pass 1 with per-wave tiles: fast parity set, count step timing, then the in-kernel clocks (ABLATE build)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
TAG=${1:-r04p1b}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --durations=5 > "$OUT/pytest.log" 2>&1 || { tail -60 "$OUT/pytest.log"; exit 1; }
tail -3 "$OUT/pytest.log"
for win in 18 16; do echo "== SHK_PART_WIN=$win"; SHK_PART_WIN=$win timeout -k 10 200 python tools/pre_only.py 2>&1 | tail -1; done
cp sparrowhawk_amd/libshk_hip_ablate.so sparrowhawk_amd/libshk_hip.so      # (on the GPU box's copy of the tree only)
for D in 0 1 2; do
  echo "== ABLATE SHK_DEBUG_NOSTORE=$D"; SHK_DEBUG_P1CLK=$([ $D = 1 ] && echo 0 || echo 1) SHK_DEBUG_NOSTORE=$D timeout -k 10 120 python3 tools/pre_only.py 2>&1 | grep -E "p1clk|partition_kernel" | tail -2
done
echo "== ABLATE no clocks"; for D in 0 2; do SHK_DEBUG_NOSTORE=$D timeout -k 10 120 python3 tools/pre_only.py 2>&1 | grep -E "partition_kernel" | tail -1; done
