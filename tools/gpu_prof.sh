#!/bin/bash
# one gpurun call: quick tests, bench under rocprofv3 kernel trace (per-kernel table), EM timing.  usage: gpu_prof.sh <tag>
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_$1
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_solver.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python3 -c "
import json
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','mttkrp_mode1_ms']}, d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/bench_prof.err; echo "prof rc=$?"
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 $R/tools/kernel_timeline.py {} 12 > $OUT/timeline.txt
find $OUT/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_trace.csv
rm -rf $OUT/prof
cat $OUT/timeline.txt
cd $R
timeout -k 10 300 python tools/time_em.py > $OUT/em.log 2>&1; cat $OUT/em.log | tail -3
