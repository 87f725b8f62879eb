#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcdbg
mkdir -p $OUT
for d in 2 20 0; do
  AOADMM_CONTRACT_DBG=$d rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $OUT/d$d -- python3 $R/tools/perf_mttkrp.py 2000 20 f32 1 > $OUT/d$d.log 2>&1
done
ls $OUT
