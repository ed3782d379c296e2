#!/bin/bash
# bf16x6 GEMM held to one workgroup per CU by its LDS size (81 KB), pooling padding 78.5 KB so that one of each fits a CU
export INFV_LTM_LIBRARY=exp
{
tools/env_sweep.sh "INFV_NONE=0" "INFV_X6_LDS=82944 INFV_PR_PAD=80384" "INFV_X6_LDS=82944 INFV_PR_PAD=80384 INFV_PR_U=4" "INFV_NONE=1" "INFV_X6_LDS=82944 INFV_PR_PAD=80384"
INFV_X6_LDS=82944 INFV_PR_PAD=80384 INFV_WG_STAMPS=1 python tools/residency.py x6l 2>&1 | grep -v amdgpu.ids | tail -20
python tools/launch_table.py gpurun_out/wg_stamps_x6l.npy 20 5
} 2>&1 | tee gpurun_out/sweep_r05c.txt
