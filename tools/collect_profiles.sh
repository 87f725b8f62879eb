#!/bin/bash
# Round-end evidence in one gpurun call: GPU suite, bench lines (fp32 incl. fp64 drift run, fp64), rocprofv3 kernel
# tables of the bench and of the small configurations, prox kernel timings, config timings.  Copies what profiles/
# keeps (small files only) under gpurun_out/r02_final/.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_final
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/gputests.log
tail -2 $OUT/gputests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/r02_bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --prec f64 --no-cpu-baseline > $OUT/r02_bench_f64.json 2> $OUT/bench_f64.err; echo "bench f64 rc=$?"
# the driver's launch shape for one rank, and the N > 1 data path (own-rows buffer + ncclAllReduce of every MTTKRP output)
# on a one-rank RCCL communicator: the only forms of the multi-GPU path a one-GPU box can run at full size
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-drift > $OUT/r02_bench_torchrun_n1.json 2> $OUT/torchrun.err; echo "torchrun n1 rc=$?"
AOADMM_BENCH_ONE_RANK_COMM=1 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-drift > $OUT/r02_bench_one_rank_comm.json 2> $OUT/onerank.err; echo "one-rank comm rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-drift > $OUT/r02_bench_under_rocprof.json 2> $OUT/prof.err; echo "prof rc=$?"
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/r02_kernel_stats.csv
find $OUT/prof -name "*kernel_trace.csv" | head -1 | xargs -I{} python3 $R/tools/kernel_timeline.py {} 12 > $OUT/r02_kernel_timeline.txt
rm -rf $OUT/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 10 --warmup 2 --prec f64 --no-cpu-baseline > $OUT/r02_bench_f64_under_rocprof.json 2> $OUT/prof64.err; echo "prof f64 rc=$?"
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/r02_kernel_stats_f64.csv
rm -rf $OUT/prof
for c in 1 3 4; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p$c -- python3 $R/tools/time_cfg$c.py > /dev/null 2>&1
  find $OUT/p$c -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/r02_cfg${c}_kernel_stats.csv
  rm -rf $OUT/p$c
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pp -- python3 $R/tools/time_prox.py > /dev/null 2>&1
find $OUT/pp -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/r02_prox_kernel_stats.csv
rm -rf $OUT/pp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pt -- python3 $R/tools/time_tv_long.py > /dev/null 2>&1
find $OUT/pt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/r02_tv_long_kernel_stats.csv
rm -rf $OUT/pt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pe -- python3 $R/tools/time_em.py > $OUT/r02_em_timing.txt 2>&1
find $OUT/pe -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/r02_em_kernel_stats.csv
rm -rf $OUT/pe
cd $R
timeout -k 10 600 python tools/time_configs.py > $OUT/r02_configs_timing.txt 2>&1
echo "--- per-configuration scripts (slope between two solve lengths) ---" >> $OUT/r02_configs_timing.txt
for c in 1 2 3 4; do timeout -k 10 300 python tools/time_cfg$c.py 2>&1 | grep -i "per outer iteration\|ms/iter" >> $OUT/r02_configs_timing.txt; done
cat $OUT/r02_configs_timing.txt
python3 -c "
import json
for f in ['r02_bench.json','r02_bench_f64.json']:
    d=json.loads(open('$OUT/'+f).read().strip().splitlines()[-1])
    print(f, {k:d.get(k) for k in ['value','ms_per_step','mttkrp_mode1_ms','replicated_tail_ms']}, d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['traffic'], d.get('fp32_drift'), d.get('tail_breakdown'))"
