#!/bin/bash
# round 3, sixth GPU call: trimmed TV kernel -- prox and solver tests, then the rank-of-8 timeline
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_c6
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_solver.py tests/test_gpu_sharded.py tests/test_gpu_fuzz.py tests/test_known_answers.py tests/test_gpu_random_models.py tests/test_golden.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --as-rank 0 --of 8 > $OUT/rank0_of_8.json 2> $OUT/rank0_of_8.err && echo "as-rank ok"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-drift > $OUT/bench.json 2> $OUT/bench.err && echo "bench ok"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 10 --warmup 2 --as-rank 0 --of 8 > $OUT/rank0_of_8_under_rocprof.json 2> $OUT/prof.err; echo "prof rc=$?"
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/rank0_of_8_kernel_stats.csv
find $OUT/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $OUT/rank0_of_8_kernel_trace.csv
rm -rf $OUT/prof
grep -E "prox_tv|admm_rows|reduce_|atb_|sys_build|contract16" $OUT/rank0_of_8_kernel_stats.csv | cut -c1-140
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    tb=d.get('tail_breakdown') or {}
    print(f.split('/')[-1], {k:round(d.get(k),4) for k in ['value','ms_per_step','mttkrp_mode1_ms','replicated_tail_ms']}, 'pass', round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4), 'small', tb.get('replicated_small_kernels_ms'), 'red', tb.get('t_reductions_ms'))
PY
