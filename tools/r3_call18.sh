#!/bin/bash
# round 3: TV kernel, 512 against 1024 threads per column at 2000 rows (kernel table at the share of 8)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_c18
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_solver.py tests/test_known_answers.py -m gpu -x -q -k "tv or TV or prox or script10" > $OUT/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 $OUT/tests.log
[ $rc -eq 0 ] || exit 1
cd /tmp && export TMPDIR=/tmp
for v in 1024 512; do
  if [ $v = 512 ]; then export AOADMM_TV_512=1; else unset AOADMM_TV_512; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 20 --warmup 2 --as-rank 0 --of 8 > $OUT/rank0_of_8_$v.json 2> /dev/null
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); echo "$v: $(grep prox_tv $f | cut -d, -f1-4 | cut -c1-90)"
  rm -rf $OUT/prof
  for rep in 1 2; do
  timeout -k 10 200 python3 $R/bench.py --steps 20 --warmup 5 --as-rank 0 --of 8 2> /dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$v', 'of 8: ms_per_step', round(d['ms_per_step'],4), 'small', round(d['tail_breakdown']['replicated_small_kernels_ms'],4))"
  done
done
