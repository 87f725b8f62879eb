#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in 310212 311219; do
echo "case $c two-level:"; python tools/fuzz_solver.py 1 $c 2>&1 | tail -3 | cut -c1-300
echo "case $c single-level:"; AOADMM_CONTRACT_FLUSH=0 python tools/fuzz_solver.py 1 $c 2>&1 | tail -3 | cut -c1-300
done
timeout -k 10 300 python -m pytest tests/test_gpu_solver.py -m gpu -x -q -k "natural_copy or em_missing_cp or cp_tv" 2>&1 | tail -3
