#!/bin/bash
# the multi-rank campaign on ONE rank count with per-rank stage logs (which case, which stage)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
W=${1:-3}; N=${2:-250}; SEED=${3:-9500}
mkdir -p gpurun_out/fzlog; rm -f gpurun_out/fzlog/*
export SHK_DIST_FUZZ_CASES=$N SHK_DIST_FUZZ_SEED=$SEED MOCK_RCCL_JITTER=1 MOCK_RCCL_SEED=$SEED SHK_DIST_FUZZ_LOG=$PWD/gpurun_out/fzlog/log SHK_STAGE_LOG=1
timeout -k 10 600 python3 -m pytest tests/test_dist.py -m gpu -x -q -k "sharded_graph_several_ranks and $W" > gpurun_out/fzlog/out.txt 2>&1
tail -3 gpurun_out/fzlog/out.txt
for r in 0 1 2 3; do [ -f gpurun_out/fzlog/log.$r ] && { echo "== rank $r"; tail -n 12 gpurun_out/fzlog/log.$r | cut -c1-300; }; done
# keep the merge-back small
for r in 0 1 2 3; do [ -f gpurun_out/fzlog/log.$r ] && { tail -n 400 gpurun_out/fzlog/log.$r > gpurun_out/fzlog/tail.$r; rm gpurun_out/fzlog/log.$r; }; done
