#!/bin/bash
# round 3: EM pass, write-back of every vector without the per-line test (update = 2) against the line test, 20 % missing
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_c22
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in line all; do
  if [ $v = all ]; then export AOADMM_EM_STORE_ALL=1; else unset AOADMM_EM_STORE_ALL; fi
  for f in 0.2; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pe -- python3 $R/tools/time_em.py $f 2>/dev/null | grep "mask=True" | sed "s/^/$v: /"
    find $OUT/pe -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/em_kernel_stats_${v}_$f.csv
    rm -rf $OUT/pe
    grep "em_cp" $OUT/em_kernel_stats_${v}_$f.csv | cut -c1-130
  done
done
