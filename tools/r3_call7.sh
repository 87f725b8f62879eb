#!/bin/bash
# round 3, seventh GPU call: TV with Pc/start/J in LDS (4097..12000 rows), rider with priority, tightened drift test
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_c7
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_solver.py tests/test_gpu_fullsize.py tests/test_gpu_sharded.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
for v in hybrid allglobal; do
  if [ $v = allglobal ]; then export AOADMM_TV_ALL_GLOBAL=1; else unset AOADMM_TV_ALL_GLOBAL; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pt -- python3 $R/tools/time_tv_long_loop.py > /dev/null 2>&1
  find $OUT/pt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/tv_long_loop_kernel_stats_$v.csv
  rm -rf $OUT/pt
  grep "prox_tv" $OUT/tv_long_loop_kernel_stats_$v.csv | cut -c1-200
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pt -- python3 $R/tools/time_tv_long.py > $OUT/tv_long_$v.txt 2>&1
  find $OUT/pt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/tv_long_kernel_stats_$v.csv
  rm -rf $OUT/pt
  grep -i "rows" $OUT/tv_long_$v.txt | head -12
done
unset AOADMM_TV_ALL_GLOBAL
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 10 --warmup 2 --as-rank 0 --of 8 > $OUT/rank0_of_8_under_rocprof.json 2> $OUT/prof.err; echo "prof rc=$?"
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/rank0_of_8_kernel_stats.csv
rm -rf $OUT/prof
grep -E "reduce_|sys_build" $OUT/rank0_of_8_kernel_stats.csv | cut -c1-60,150-260
cd $R
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --as-rank 0 --of 8 > $OUT/rank0_of_8.json 2> $OUT/rank0_of_8.err && python3 -c "
import json; d=json.loads(open('$OUT/rank0_of_8.json').read().strip().splitlines()[-1]); print('rank0_of_8', d['ms_per_step'], d['tail_breakdown'])"
