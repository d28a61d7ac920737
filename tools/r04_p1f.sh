#!/bin/bash
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
for i in 1 2; do timeout -k 10 200 python tools/pre_only.py 2>&1 | tail -1 | cut -c1-120; done
echo "== K=51 masked"; K=51 ERR=0.01 MASK=1 timeout -k 10 200 python tools/pre_only.py 2>&1 | tail -1 | cut -c1-120
echo "== K=51 errors"; K=51 ERR=0.01 timeout -k 10 200 python tools/pre_only.py 2>&1 | tail -1 | cut -c1-120
