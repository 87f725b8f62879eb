#!/bin/bash
# round 3, first GPU call: new tests + two-level accumulation (time, drift) + one rank's share of an 8-GPU job
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_c1
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_known_answers.py tests/test_gpu_sharded.py -m gpu -x -q > $OUT/tests_a.log 2>&1; echo "tests_a rc=$?"; tail -3 $OUT/tests_a.log
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_l2.json 2> $OUT/bench_l2.err; echo "bench l2 rc=$?"
AOADMM_CONTRACT_FLUSH=0 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_l1.json 2> $OUT/bench_l1.err; echo "bench l1 rc=$?"
for f in 1 2 6; do
AOADMM_CONTRACT_FLUSH=$f timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_flush$f.json 2> $OUT/bench_flush$f.err; echo "bench flush $f rc=$?"
done
for n in 8 4 2; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --as-rank 0 --of $n > $OUT/rank0_of_$n.json 2> $OUT/rank0_of_$n.err; echo "as-rank of $n rc=$?"
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --as-rank 3 --of 8 > $OUT/rank3_of_8.json 2> $OUT/rank3_of_8.err; echo "as-rank 3 of 8 rc=$?"
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    dr=d.get('fp32_drift') or {}
    pm=d.get('parity_mode') or {}
    print(f.split('/')[-1], {k:round(d.get(k),4) for k in ['value','ms_per_step','mttkrp_mode1_ms','replicated_tail_ms']}, 'pass', round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4), 'drift', dr.get('factor_rel_fro_f32_vs_f64'), 'tail', (d.get('tail_breakdown') or {}).get('replicated_small_kernels_ms'), (d.get('tail_breakdown') or {}).get('t_reductions_ms'), 'parity', pm.get('ms_per_step'), (pm.get('roofline') or {}).get('frac'))
PY
