#!/bin/bash
# one gpurun call: GPU test suite, fp32 and fp64 bench lines.  usage: gpu_check.sh <tag> [notests]
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_$1
mkdir -p $OUT
cd $R
if [ "${2:-}" != "notests" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/gputests.log
  tail -3 $OUT/gputests.log
fi
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python3 -c "
import json,sys
d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','mttkrp_mode1_ms']}, d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --prec f64 --no-cpu-baseline > $OUT/bench_f64.json 2> $OUT/bench_f64.err; echo "bench f64 rc=$?"
python3 -c "
import json,sys
d=json.loads(open('$OUT/bench_f64.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ['value','ms_per_step','mttkrp_mode1_ms']}, d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
