#!/bin/bash
# does the ragged last round of workgroups cost what the model says?  pass GB/s at cube sizes just below / above a whole number of rounds
set -u
cd $GRAFT_REPO_ROOT
for n in 1980 1984 2000 2048; do echo "size $n  (workgroups/512 = $(python3 -c "print(round(((($n+3)//4*4)*$n+511)//512/512,3))"))"; timeout -k 10 200 python tools/perf_mttkrp.py $n 20 f32 5 2>&1 | grep "mode"; done
