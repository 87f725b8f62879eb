#!/bin/bash
# one gpurun call: GPU test suite, bench line, rocprofv3 kernel trace of the bench
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_base
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/gputests.log
tail -3 $OUT/gputests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/bench_prof.err; echo "prof rc=$?"
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 $R/tools/kernel_timeline.py {} 12 > $OUT/timeline.txt
find $OUT/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_trace.csv
rm -rf $OUT/prof
cat $OUT/timeline.txt
