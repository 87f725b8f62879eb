#!/bin/bash
# round 3, third GPU call: whole GPU suite; single memset + system-build rider (on / off) at one rank of 8 and at N = 1;
# config 4 sharded over 4 ranks vs repeated
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_c3
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/gputests.log
for n in 8; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --as-rank 0 --of $n > $OUT/rank0_of_${n}.json 2> $OUT/rank0_of_$n.err; echo "as-rank of $n rc=$?"
AOADMM_NO_SYS_RIDER=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --as-rank 0 --of $n > $OUT/rank0_of_${n}_norider.json 2> $OUT/rank0_of_${n}_norider.err; echo "as-rank of $n norider rc=$?"
done
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-drift > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
AOADMM_NO_SYS_RIDER=1 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-drift > $OUT/bench_norider.json 2> $OUT/bench_norider.err; echo "bench norider rc=$?"
timeout -k 10 300 python tools/time_cfg4.py > $OUT/cfg4_single.txt 2>&1; echo "cfg4 single rc=$?"
timeout -k 10 300 python tools/time_cfg4.py --sharded 4 > $OUT/cfg4_sharded4.txt 2>&1; echo "cfg4 sharded rc=$?"
timeout -k 10 300 python tools/time_cfg4.py --as-rank 0 --of 4 > $OUT/cfg4_rank0_of_4.txt 2>&1; echo "cfg4 share rc=$?"
cat $OUT/cfg4_single.txt $OUT/cfg4_sharded4.txt $OUT/cfg4_rank0_of_4.txt | grep -v "^cfg4 K=256: [0-9]* outer"
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
