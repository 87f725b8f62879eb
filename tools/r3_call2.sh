#!/bin/bash
# round 3: kernel table + timeline of one rank's share of the 8-GPU job
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_c2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 10 --warmup 2 --as-rank 0 --of 8 > $OUT/rank0_of_8_under_rocprof.json 2> $OUT/prof.err; echo "prof rc=$?"
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/rank0_of_8_kernel_stats.csv
find $OUT/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 $R/tools/kernel_timeline.py {} 12 > $OUT/rank0_of_8_kernel_timeline.txt
find $OUT/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $OUT/rank0_of_8_kernel_trace.csv
rm -rf $OUT/prof
head -50 $OUT/rank0_of_8_kernel_timeline.txt
