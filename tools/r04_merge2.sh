#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
for rep in 1 2 3; do for M in 1 2; do
  echo "MERGE=$M $(SHK_COUNT_MERGE=$M timeout -k 10 200 python tools/step_breakdown.py 2>&1 | grep -E "^(preprocess|assemble|sum)" | tr '\n' ' ')"
done; done
