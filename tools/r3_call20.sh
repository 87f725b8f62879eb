#!/bin/bash
# round 3: TV on a 6000-row mode in the ADMM loop after "merges and splits in one round" (kernel table), + TV / prox tests
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_c20
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_solver.py tests/test_known_answers.py -m gpu -x -q -k "tv or TV or prox or script10" > $OUT/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 $OUT/tests.log
[ $rc -eq 0 ] || exit 1
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pt -- python3 $R/tools/time_tv_long_loop.py > /dev/null 2>&1
find $OUT/pt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/tv_long_loop_kernel_stats_$rep.csv
rm -rf $OUT/pt
grep "prox_tv" $OUT/tv_long_loop_kernel_stats_$rep.csv | cut -c1-160
done
