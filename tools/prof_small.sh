#!/bin/bash
# rocprofv3 kernel tables of the small BASELINE configurations (1, 3, 4).  usage: prof_small.sh <tag>
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in 1 3 4; do
  python3 $R/tools/time_cfg$c.py 2>&1 | grep "per outer iteration"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p$c -- python3 $R/tools/time_cfg$c.py > /dev/null 2>&1
  find $OUT/p$c -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/cfg${c}_kernel_stats.csv
  rm -rf $OUT/p$c
done
