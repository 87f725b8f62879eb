#!/bin/bash
# round 3: row solve inside the TV kernel (one launch per inner iteration from iteration 1 on) -- tests, then one rank's share with / without
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_c14
mkdir -p $OUT
cd $R
timeout -k 10 700 python -m pytest tests/test_gpu_solver.py tests/test_known_answers.py tests/test_gpu_fuzz.py tests/test_gpu_random_models.py tests/test_gpu_sharded.py -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $OUT/tests.log
[ $rc -eq 0 ] || exit 1
for v in solve pair; do
  if [ $v = pair ]; then export AOADMM_NO_TV_SOLVE=1; else unset AOADMM_NO_TV_SOLVE; fi
  for rep in 1 2; do
    timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --as-rank 0 --of 8 > $OUT/rank0_of_8_${v}_$rep.json 2> /dev/null || exit 1
    python3 -c "import json;d=json.loads(open('$OUT/rank0_of_8_${v}_$rep.json').read().strip().splitlines()[-1]);print('$v', $rep, 'of 8: ms_per_step', round(d['ms_per_step'],4), d.get('tail_breakdown'))"
  done
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-drift > $OUT/bench_${v}.json 2> /dev/null || exit 1
  python3 -c "import json;d=json.loads(open('$OUT/bench_${v}.json').read().strip().splitlines()[-1]);print('$v', 'N=1: ms_per_step', round(d['ms_per_step'],4), d.get('tail_breakdown'))"
done
unset AOADMM_NO_TV_SOLVE
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 10 --warmup 2 --as-rank 0 --of 8 > $OUT/rank0_of_8_under_rocprof.json 2> /dev/null
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); python3 $R/tools/iteration_sequence.py $f 6 9 > $OUT/iteration_sequence.txt; tail -62 $OUT/iteration_sequence.txt | cut -c1-80
rm -rf $OUT/prof
