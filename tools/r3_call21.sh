#!/bin/bash
# round 3: TV kernel at 25 bytes of LDS per row (all-LDS up to 6400 rows) -- full GPU suite, share of 8, headline
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_c21
mkdir -p $OUT
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $OUT/tests.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2; do
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --as-rank 0 --of 8 > $OUT/rank0_of_8_$rep.json 2> /dev/null || exit 1
  python3 -c "import json;d=json.loads(open('$OUT/rank0_of_8_$rep.json').read().strip().splitlines()[-1]);print($rep, 'of 8: ms_per_step', round(d['ms_per_step'],4), 'small', round(d['tail_breakdown']['replicated_small_kernels_ms'],4), 'passes', round(d['tail_breakdown']['tensor_passes_ms'],4))"
done
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-drift > $OUT/bench.json 2> /dev/null || exit 1
python3 -c "import json;d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]);print('N=1: ms_per_step', round(d['ms_per_step'],4), 'small', round(d['tail_breakdown']['replicated_small_kernels_ms'],4))"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 10 --warmup 2 --as-rank 0 --of 8 > $OUT/rank0_of_8_under_rocprof.json 2> /dev/null
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); grep "prox_tv\|admm_rows_k" $f | cut -d, -f1-4 | cut -c1-50,100-200
rm -rf $OUT/prof
