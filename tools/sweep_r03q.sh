#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}
tools/env_sweep.sh "INFV_NONE=0" "INFV_GEMM_LW=0" "INFV_NONE=1" "INFV_GEMM_LW=0 INFV_POOL_UNROLL=8" "INFV_GEMM_LW=0 INFV_POOL_PAD=60000" 2>&1 | tee gpurun_out/sweep_r03q.txt
