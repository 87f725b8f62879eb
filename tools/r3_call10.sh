#!/bin/bash
# round 3, tenth GPU call: rider in the small one-launch MTTKRP -- whole GPU suite, configs 1-4
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_c10
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.log
for c in 1 3 4; do
  timeout -k 10 200 python tools/time_cfg$c.py 2>&1 | grep -i "per outer iteration"
  AOADMM_NO_SYS_RIDER=1 timeout -k 10 200 python tools/time_cfg$c.py 2>&1 | grep -i "per outer iteration" | sed 's/^/  (system build in its own launch) /'
done
