#!/bin/bash
# round 3, fifth GPU call: one-launch element-wise ADMM loop (grid barrier): tests, then N = 1 and rank-of-8 timing on / off
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_c5
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_ops.py tests/test_gpu_sharded.py tests/test_gpu_fuzz.py tests/test_known_answers.py tests/test_gpu_random_models.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --as-rank 0 --of 8 > $OUT/rank0_of_8.json 2> $OUT/rank0_of_8.err && echo "as-rank ok"
AOADMM_TWO_LAUNCH_LOOP=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --as-rank 0 --of 8 > $OUT/rank0_of_8_two.json 2> $OUT/rank0_of_8_two.err && echo "as-rank two ok"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-drift > $OUT/bench.json 2> $OUT/bench.err && echo "bench ok"
AOADMM_TWO_LAUNCH_LOOP=1 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-drift > $OUT/bench_two.json 2> $OUT/bench_two.err && echo "bench two ok"
for c in 1 2 3; do timeout -k 10 200 python tools/time_cfg$c.py 2>&1 | grep -i "per outer iteration\|ms/iter"; done
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
