#!/bin/bash
# PMC passes for the contraction kernel (separate runs per counter group; see MI355X_MICROARCH.md)
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/tools/perf_mttkrp.py 2000 20 ${PREC:-f32} 2 > $OUT/$name.log 2>&1; }
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
ls $OUT/*
