#!/bin/bash
# re-runs single fuzz cases with the two-level accumulation of the fp32 pass on and off (usage: r3_fuzz_cases.sh <case> ...)
cd $GRAFT_REPO_ROOT
for c in "$@"; do
echo "case $c two-level:"; python tools/fuzz_solver.py 1 $c 2>&1 | tail -3 | cut -c1-300
echo "case $c single-level:"; AOADMM_CONTRACT_FLUSH=0 python tools/fuzz_solver.py 1 $c 2>&1 | tail -3 | cut -c1-300
done
