#!/bin/bash
# round 3: TV kernel with the row solve + merges and splits in one round -- full GPU suite, then the share of 8 and one GPU
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_c17
mkdir -p $OUT
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $OUT/tests.log
[ $rc -eq 0 ] || exit 1
for v in solve pair; do
  if [ $v = pair ]; then export AOADMM_NO_TV_SOLVE=1; else unset AOADMM_NO_TV_SOLVE; fi
  for rep in 1 2; do
    timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --as-rank 0 --of 8 > $OUT/rank0_of_8_${v}_$rep.json 2> /dev/null || exit 1
    python3 -c "import json;d=json.loads(open('$OUT/rank0_of_8_${v}_$rep.json').read().strip().splitlines()[-1]);print('$v', $rep, 'of 8: ms_per_step', round(d['ms_per_step'],4), 'small', round(d['tail_breakdown']['replicated_small_kernels_ms'],4), 'passes', round(d['tail_breakdown']['tensor_passes_ms'],4))"
  done
done
unset AOADMM_NO_TV_SOLVE
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-drift > $OUT/bench.json 2> /dev/null || exit 1
python3 -c "import json;d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]);print('N=1: ms_per_step', round(d['ms_per_step'],4), d.get('tail_breakdown'))"
