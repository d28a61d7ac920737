# Random small graphs (errors, repeats, hairpins, plasmids, tandem rings) through the SHARDED path on 2, 3 and 4 ranks of the
# one GPU (bytes through tests/mock_rccl), records sent raw / deduplicated in turn: every rank must end with the oracle's bytes.
# Usage (via gpurun): bash tools/fuzz_sharded.sh [cases per world] [seed] [jitter: 1 = the stand-in transport at its most hostile on every rank count]
set -e
mkdir -p gpurun_out/fuzz_sharded
export SHK_DIST_FUZZ_CASES=${1:-200} SHK_DIST_FUZZ_SEED=${2:-9100}
if [ -n "${3:-}" ]; then export MOCK_RCCL_JITTER=$3 MOCK_RCCL_SEED=${2:-9100}; fi
timeout -k 10 1100 python3 -m pytest tests/test_dist.py -m gpu -x -q -k "sharded_graph_several_ranks" 2>&1 | tee gpurun_out/fuzz_sharded/out.txt | tail -n 3
