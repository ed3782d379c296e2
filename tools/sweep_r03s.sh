#!/bin/bash
# alternating A/B on one box: default pooling (pool_frames + build_rows) vs pool_rows2_kernel with 8-load bursts
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}
tools/env_sweep.sh "INFV_NONE=0" "INFV_POOL_ROWS=2 INFV_PR_U=8" "INFV_NONE=1" "INFV_POOL_ROWS=2 INFV_PR_U=8" "INFV_NONE=2" "INFV_POOL_ROWS=2 INFV_PR_U=8" "INFV_NONE=3" "INFV_POOL_ROWS=2 INFV_PR_U=8" "INFV_POOL_ROWS=2 INFV_PR_U=4" "INFV_POOL_ROWS=2 INFV_PR_U=4" 2>&1 | tee gpurun_out/sweep_r03s.txt
