#!/bin/bash
# the multi-rank campaign against ANOTHER build of the library (sparrowhawk_amd/<name>), e.g. the one of an earlier commit
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
LIB=${1:?}; W=${2:-3}; N=${3:-250}; SEED=${4:-9500}
cp sparrowhawk_amd/$LIB sparrowhawk_amd/libshk_hip.so      # (on the GPU box's copy of the tree only)
mkdir -p gpurun_out/fzold
export SHK_DIST_FUZZ_CASES=$N SHK_DIST_FUZZ_SEED=$SEED MOCK_RCCL_JITTER=1 MOCK_RCCL_SEED=$SEED
timeout -k 10 600 python3 -m pytest tests/test_dist.py -m gpu -x -q -k "sharded_graph_several_ranks and $W" > gpurun_out/fzold/out_$LIB.txt 2>&1
tail -3 gpurun_out/fzold/out_$LIB.txt; grep -n "Memory access fault\|rror:" gpurun_out/fzold/out_$LIB.txt | head -5
