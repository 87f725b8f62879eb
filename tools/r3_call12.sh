#!/bin/bash
# waves per workgroup of the fp32 pass: 4 vs 2, at one rank's share of 8 / 4 / 2 GPUs and at one GPU (rocprofv3 kernel table)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03_c12
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in 4 2; do
  export AOADMM_CONTRACT_WPW=$w
  for n in 8 4; do
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 10 --warmup 2 --as-rank 0 --of $n > $OUT/rank0_of_${n}_wpw$w.json 2> /dev/null
    f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); echo "wpw=$w of $n: $(grep contract16 $f | cut -d, -f2-4) ms_per_step=$(python3 -c "import json;print(round(json.loads(open('$OUT/rank0_of_${n}_wpw$w.json').read().strip().splitlines()[-1])['ms_per_step'],4))")"
    rm -rf $OUT/prof
  done
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-drift > $OUT/bench_wpw$w.json 2> /dev/null
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); echo "wpw=$w N=1: $(grep contract16 $f | cut -d, -f2-4) ms_per_step=$(python3 -c "import json;print(round(json.loads(open('$OUT/bench_wpw$w.json').read().strip().splitlines()[-1])['ms_per_step'],4))")"
  rm -rf $OUT/prof
done
unset AOADMM_CONTRACT_WPW
cd $R && timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "mttkrp" 2>&1 | tail -2
