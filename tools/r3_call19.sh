#!/bin/bash
# round 3: tail split of the tensor pass -- tests, then the share of 8 / 4 / 2 and one GPU with and without
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_c19
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_ops.py tests/test_gpu_sharded.py tests/test_gpu_fullsize.py -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $OUT/tests.log
[ $rc -eq 0 ] || exit 1
for v in split nosplit; do
  if [ $v = nosplit ]; then export AOADMM_CONTRACT_TAIL=0; else unset AOADMM_CONTRACT_TAIL; fi
  for n in 8 4 2; do
    timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --as-rank 0 --of $n > $OUT/rank0_of_${n}_$v.json 2> /dev/null || exit 1
    python3 -c "import json;d=json.loads(open('$OUT/rank0_of_${n}_$v.json').read().strip().splitlines()[-1]);print('$v of $n: ms_per_step', round(d['ms_per_step'],4), 'passes', round(d['tail_breakdown']['tensor_passes_ms'],4), 'red', round(d['tail_breakdown']['t_reductions_ms'],4), 'small', round(d['tail_breakdown']['replicated_small_kernels_ms'],4), 'pass avg ms', round(d['roofline']['avg_launch_ms'],4))"
  done
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-drift > $OUT/bench_$v.json 2> /dev/null || exit 1
  python3 -c "import json;d=json.loads(open('$OUT/bench_$v.json').read().strip().splitlines()[-1]);print('$v N=1: ms_per_step', round(d['ms_per_step'],4), 'passes', round(d['tail_breakdown']['tensor_passes_ms'],4), 'red', round(d['tail_breakdown']['t_reductions_ms'],4), 'pass avg ms', round(d['roofline']['avg_launch_ms'],4), 'parity_mode', d.get('parity_mode',{}).get('ms_per_step'))"
done
