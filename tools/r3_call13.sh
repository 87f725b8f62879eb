#!/bin/bash
# round 3: EM pass on the matrix cores -- EM tests, then timing against the vector-pipe form
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_c13
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_solver.py tests/test_gpu_sharded.py tests/test_known_answers.py tests/test_gpu_fuzz.py tests/test_gpu_random_models.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -5 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
for v in mfma valu; do
  if [ $v = valu ]; then export AOADMM_EM_VALU=1; else unset AOADMM_EM_VALU; fi
  for f in 0.2 0.01; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pe -- python3 $R/tools/time_em.py $f 2>/dev/null | grep "mask=True" | sed "s/^/$v: /"
    find $OUT/pe -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/em_kernel_stats_${v}_$f.csv
    rm -rf $OUT/pe
    grep "em_cp" $OUT/em_kernel_stats_${v}_$f.csv | cut -c1-130
  done
done
