#!/bin/bash
# PMC summaries of the two streaming contraction kernels for this round (separate rocprofv3 --pmc passes each)
set -u
R=$GRAFT_REPO_ROOT
for PREC in f32 f64; do
  export PREC
  rm -rf $R/gpurun_out/pmc
  bash $R/tools/pmc_contract.sh > /dev/null 2>&1
  if [ "$PREC" = f32 ]; then K=contract16_f32; ALGO=32320000000; else K=contract_f64; ALGO=64640000000; fi
  python3 $R/tools/pmc_summarize.py $R/gpurun_out/pmc $K $ALGO $R/gpurun_out/r02_pmc_$K.json
  rm -rf $R/gpurun_out/pmc/*/
done
