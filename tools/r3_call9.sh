#!/bin/bash
# round 3, ninth GPU call: register-resident PARAFAC2 slab iteration -- whole GPU suite, then configs 1 and 4 with / without
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_c9
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.log
for c in 1 4; do
  timeout -k 10 200 python tools/time_cfg$c.py 2>&1 | grep -i "per outer iteration"
  AOADMM_NO_PAR2_REGS=1 timeout -k 10 200 python tools/time_cfg$c.py 2>&1 | grep -i "per outer iteration" | sed 's/^/  (LDS form) /'
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p4 -- python3 $R/tools/time_cfg4.py > /dev/null 2>&1
find $OUT/p4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/cfg4_kernel_stats.csv
rm -rf $OUT/p4
head -8 $OUT/cfg4_kernel_stats.csv | cut -c1-150
