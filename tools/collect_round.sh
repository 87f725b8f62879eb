#!/bin/bash
# Round-end evidence in one gpurun call: GPU suite, bench lines (fp32 incl. fp64 drift / parity-mode run, fp64, one rank's share
# of 2/4/8 GPUs), rocprofv3 kernel tables of the bench, of the rank-of-8 share and of the small configurations, prox / EM
# kernel timings, config timings, PMC passes.  Copies what profiles/ keeps (small files only) under gpurun_out/<tag>_final/.
# usage: collect_round.sh <tag, e.g. r03> [notests]
set -u
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${TAG}_final
mkdir -p $OUT
cd $R
if [ "${2:-}" != "notests" ]; then
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/gputests.log
tail -2 $OUT/gputests.log
fi
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --prec f64 --no-cpu-baseline > $OUT/${TAG}_bench_f64.json 2> $OUT/bench_f64.err; echo "bench f64 rc=$?"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-drift > $OUT/${TAG}_bench_torchrun_n1.json 2> $OUT/torchrun.err; echo "torchrun n1 rc=$?"
for n in 8 4 2; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --as-rank 0 --of $n > $OUT/${TAG}_rank0_of_$n.json 2> $OUT/rank0_of_$n.err; echo "rank 0 of $n rc=$?"
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --as-rank 7 --of 8 > $OUT/${TAG}_rank7_of_8.json 2> $OUT/rank7_of_8.err; echo "rank 7 of 8 rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-drift > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/prof.err; echo "prof rc=$?"
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_kernel_stats.csv
find $OUT/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 $R/tools/kernel_timeline.py {} 12 > $OUT/${TAG}_kernel_timeline.txt
rm -rf $OUT/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 10 --warmup 2 --prec f64 --no-cpu-baseline > $OUT/${TAG}_bench_f64_under_rocprof.json 2> $OUT/prof64.err; echo "prof f64 rc=$?"
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_kernel_stats_f64.csv
rm -rf $OUT/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 10 --warmup 2 --as-rank 0 --of 8 > $OUT/${TAG}_rank0_of_8_under_rocprof.json 2> $OUT/prof8.err; echo "prof rank-of-8 rc=$?"
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_rank0_of_8_kernel_stats.csv
find $OUT/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 $R/tools/kernel_timeline.py {} 12 > $OUT/${TAG}_rank0_of_8_kernel_timeline.txt
find $OUT/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 $R/tools/iteration_sequence.py {} 9 12 > $OUT/${TAG}_rank0_of_8_iteration_sequence.txt
rm -rf $OUT/prof
for c in 1 2 3 4; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p$c -- python3 $R/tools/time_cfg$c.py > /dev/null 2>&1
  find $OUT/p$c -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_cfg${c}_kernel_stats.csv
  rm -rf $OUT/p$c
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pp -- python3 $R/tools/time_prox.py > /dev/null 2>&1
find $OUT/pp -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_prox_kernel_stats.csv
rm -rf $OUT/pp
for f in 0.2 0.01; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pe -- python3 $R/tools/time_em.py $f 2> /dev/null | grep "mask=" > $OUT/${TAG}_em_timing_$f.txt
find $OUT/pe -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_em_kernel_stats_$f.csv
rm -rf $OUT/pe
done
cd $R
timeout -k 10 600 python tools/time_configs.py > $OUT/${TAG}_configs_timing.txt 2>&1
echo "--- per-configuration scripts (slope between two solve lengths) ---" >> $OUT/${TAG}_configs_timing.txt
for c in 1 2 3 4; do timeout -k 10 300 python tools/time_cfg$c.py 2>&1 | grep -i "per outer iteration\|ms/iter" >> $OUT/${TAG}_configs_timing.txt; done
echo "--- config 4 as written: K = 256 slabs over 4 ranks ---" >> $OUT/${TAG}_configs_timing.txt
timeout -k 10 300 python tools/time_cfg4.py --sharded 4 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" >> $OUT/${TAG}_configs_timing.txt
timeout -k 10 300 python tools/time_cfg4.py --as-rank 0 --of 4 2>&1 | grep "cfg4" >> $OUT/${TAG}_configs_timing.txt
cat $OUT/${TAG}_configs_timing.txt
# PMC passes (separate runs per counter): the two contraction kernels, then the EM kernel
for PREC in f32 f64; do
  export PREC
  rm -rf $R/gpurun_out/pmc
  bash $R/tools/pmc_contract.sh > /dev/null 2>&1
  if [ "$PREC" = f32 ]; then K=contract16_f32; ALGO=32320000000; else K=contract_f64; ALGO=64640000000; fi
  python3 $R/tools/pmc_summarize.py $R/gpurun_out/pmc $K $ALGO $OUT/${TAG}_pmc_$K.json
  rm -rf $R/gpurun_out/pmc/*/
done
bash $R/tools/pmc_em.sh $TAG > $OUT/pmc_em.log 2>&1; cp $R/gpurun_out/${TAG}_pmc_em_cp_vec.json $OUT/ 2>/dev/null
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/${TAG}_*.json')):
    if 'pmc' in f: continue
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'unreadable', e); continue
    tb=d.get('tail_breakdown') or {}
    pm=d.get('parity_mode') or {}
    print(f.split('/')[-1], {k:round(d.get(k),4) for k in ['value','ms_per_step','mttkrp_mode1_ms','replicated_tail_ms']}, 'pass', round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4), d['roofline'].get('traffic'), 'small', tb.get('replicated_small_kernels_ms'), 'red', tb.get('t_reductions_ms'), 'drift', (d.get('fp32_drift') or {}).get('factor_rel_fro_f32_vs_f64'), 'parity', pm.get('ms_per_step'), (pm.get('roofline') or {}).get('frac'), (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('extrapolated'))
PY
