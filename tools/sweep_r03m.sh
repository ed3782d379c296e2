#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # the INFV_* knobs below only exist in the experiments build (csrc/knobs.h)
# who slows the pooling stream (16-row chain tiles, V' projection on the side stream): INFV_SKIP 2 = no GEMM, 4 = no UC (+alpha), 8 = no chain
tools/env_sweep.sh "INFV_SKIP=0" "INFV_SKIP=2" "INFV_SKIP=4" "INFV_SKIP=6" "INFV_SKIP=8" "INFV_SKIP=10" "INFV_SKIP=12" "INFV_SKIP=14" \
  "INFV_SKIP=6 INFV_POOL_UNROLL=16" "INFV_SKIP=14 INFV_POOL_UNROLL=16" 2>&1 | tee gpurun_out/sweep_r03m.txt
