#!/bin/bash
# round 4: partitions x tables of pass 2 on the bench workload (shipped build)
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
for P in 2048 4096 8192; do for M in 1 2 4; do
  echo "P=$P MERGE=$M $(SHK_PART_P=$P SHK_COUNT_MERGE=$M timeout -k 10 120 python3 tools/pre_only.py 2>&1 | tail -1 | cut -c1-150)"
done; done
