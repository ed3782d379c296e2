#!/bin/bash
# pool_rows2_kernel (fused pool + rows, short-lived workgroups) against the default two-kernel pooling, same box
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}
{
INFV_POOL_ROWS=2 tools/quick_bench.sh pr2 6
tools/env_sweep.sh "INFV_NONE=0" "INFV_POOL_ROWS=2" "INFV_POOL_ROWS=2 INFV_PR_U=8" "INFV_POOL_ROWS=2 INFV_PR_PAD=60000" "INFV_POOL_ROWS=2 INFV_PR_U=2" "INFV_NONE=1"
} 2>&1 | tee gpurun_out/sweep_r03r.txt
