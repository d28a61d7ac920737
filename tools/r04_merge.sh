#!/bin/bash
# round 4: partitions per table of k_count_weighted (SHK_COUNT_MERGE) on the bench workload, alternating
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
for rep in 1 2 3; do for M in 1 2; do
  echo "MERGE=$M $(SHK_COUNT_MERGE=$M timeout -k 10 120 python3 tools/pre_only.py 2>&1 | tail -1 | cut -c1-60)"
done; done
echo "== K=51 masked"; for M in 1 2; do echo "MERGE=$M $(K=51 ERR=0.01 MASK=1 SHK_COUNT_MERGE=$M timeout -k 10 200 python tools/pre_only.py 2>&1 | tail -1 | cut -c1-60)"; done
