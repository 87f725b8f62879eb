#!/bin/bash
# round 3: EM fused pass with 8-byte vectors (AOADMM_EM_VEC=2) against the 16-byte form; EM tests both ways
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_c11
mkdir -p $OUT
cd $R
AOADMM_EM_VEC=2 timeout -k 10 600 python -m pytest tests/test_gpu_solver.py tests/test_gpu_sharded.py tests/test_known_answers.py -m gpu -x -q -k "em or missing or script12" > $OUT/tests_vec2.log 2>&1; echo "tests vec2 rc=$?"; tail -3 $OUT/tests_vec2.log
cd /tmp && export TMPDIR=/tmp
for v in 4 2; do
  export AOADMM_EM_VEC=$v
  for f in 0.2 0.01; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pe -- python3 $R/tools/time_em.py $f > $OUT/em_timing_v${v}_$f.txt 2>&1
    find $OUT/pe -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/em_kernel_stats_v${v}_$f.csv
    rm -rf $OUT/pe
    echo "VEC=$v missing=$f: $(grep 'mask=True' $OUT/em_timing_v${v}_$f.txt | tail -1)"
    grep "em_cp_vec_k" $OUT/em_kernel_stats_v${v}_$f.csv | cut -c1-140
  done
done
