#!/bin/bash
# randomised parity sweeps with fresh seeds (usage: r3_fuzz.sh <n_solver> <n_coupled> <seed>)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_fuzz
(timeout -k 10 520 python tools/fuzz_solver.py $1 $3 > gpurun_out/r03_fuzz/solver_$3.log 2>&1; echo "solver rc=$?") &
P1=$!
(timeout -k 10 520 python tools/fuzz_coupled.py $2 $3 > gpurun_out/r03_fuzz/coupled_$3.log 2>&1; echo "coupled rc=$?") &
P2=$!
while kill -0 $P1 2>/dev/null || kill -0 $P2 2>/dev/null; do sleep 60; tail -1 gpurun_out/r03_fuzz/solver_$3.log; tail -1 gpurun_out/r03_fuzz/coupled_$3.log; done
wait
grep -h "CASE" gpurun_out/r03_fuzz/solver_$3.log gpurun_out/r03_fuzz/coupled_$3.log | head -20
tail -1 gpurun_out/r03_fuzz/solver_$3.log; tail -1 gpurun_out/r03_fuzz/coupled_$3.log
