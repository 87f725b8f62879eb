#!/bin/bash
# round 3, fourth GPU call: EM with the packed mask (tests, timing at 20 % / 1 % missing, kernel table, PMC), sharded tests,
# rider-first + own-rows buffers at one rank of 8
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_c4
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_sharded.py tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --as-rank 0 --of 8 > $OUT/rank0_of_8.json 2> $OUT/rank0_of_8.err; echo "as-rank rc=$?"
timeout -k 10 300 python tools/time_em.py 0.2 > $OUT/em_timing_20.txt 2>&1; echo "em 20 rc=$?"; tail -2 $OUT/em_timing_20.txt
timeout -k 10 300 python tools/time_em.py 0.01 > $OUT/em_timing_01.txt 2>&1; echo "em 1 rc=$?"; tail -2 $OUT/em_timing_01.txt
cd /tmp && export TMPDIR=/tmp
for f in 0.2 0.01; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pe -- python3 $R/tools/time_em.py $f > /dev/null 2>&1
find $OUT/pe -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/em_kernel_stats_$f.csv
rm -rf $OUT/pe
grep "em_cp_vec" $OUT/em_kernel_stats_$f.csv
done
cd $R
bash tools/pmc_em.sh r03 > $OUT/pmc_em.log 2>&1; echo "pmc rc=$?"; tail -30 $OUT/pmc_em.log
python3 - <<PY
import json
d=json.loads(open('$OUT/rank0_of_8.json').read().strip().splitlines()[-1])
tb=d.get('tail_breakdown') or {}
print('rank0_of_8', {k:round(d.get(k),4) for k in ['value','ms_per_step','mttkrp_mode1_ms','replicated_tail_ms']}, 'pass', round(d['roofline']['avg_launch_ms'],4), 'small', tb.get('replicated_small_kernels_ms'), 'red', tb.get('t_reductions_ms'))
PY
