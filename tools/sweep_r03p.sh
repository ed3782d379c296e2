#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}
# pooling + projection GEMM only (no role S, no UC): how do the two share CUs?
export INFV_SKIP=12
tools/env_sweep.sh "INFV_NONE=0" \
 "INFV_GEMM_LW=0" \
 "INFV_POOL_UNROLL=16" \
 "INFV_POOL_UNROLL=2" \
 "INFV_POOL_NT=1024" \
 "INFV_POOL_PAD=1 INFV_POOL_UNROLL=4" \
 "INFV_POOL_PAD=1 INFV_POOL_UNROLL=16" \
 "INFV_GEMM_PAD=90000" \
 "INFV_GEMM_LW=0 INFV_GEMM_PAD=90000" \
 "INFV_GEMM_LW=0 INFV_POOL_PAD=1 INFV_POOL_UNROLL=8" \
 2>&1 | tee gpurun_out/sweep_r03p.txt
