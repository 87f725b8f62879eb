#!/bin/bash
# PMC passes over the placement probe (one contraction pass per 4 GB piece of a 96 GB allocation)
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_pmcplace
mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- $R/tools/micro/pass_placement pmc > $OUT/$name.log 2>&1; tail -1 $OUT/$name.log; }
run utcl TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum
run utcl2 TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum
run lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run tcc TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
run sq SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES
run grbm GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCC_BUSY_sum TCC_REQ_sum
for d in utcl utcl2 lat tcc ta sq grbm; do f=$(find $OUT/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $OUT/$d.csv; rm -rf $OUT/$d; done
ls -la $OUT
