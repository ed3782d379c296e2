#!/bin/bash
# the GEMM's matrix waves idle in s_nop between MFMAs (X=8: 32 cycles, X=16: 48) so that co-resident waves get the VALU port
export INFV_LTM_LIBRARY=exp INFV_PR_U=4
{
for x in 0 8 16; do echo "== pooling + GEMM only, INFV_GEMM_X=$x"; INFV_WG_STAMPS=1 INFV_SKIP=12 INFV_GEMM_X=$x python tools/residency.py x 2>&1 | grep -E "pool |gemm " ; done
tools/env_sweep.sh "INFV_NONE=0" "INFV_GEMM_X=8" "INFV_GEMM_X=16" "INFV_NONE=1" "INFV_GEMM_X=8" "INFV_GEMM_X=16"
INFV_WG_STAMPS=1 INFV_GEMM_X=16 python tools/residency.py x16 2>&1 | grep -v amdgpu.ids | tail -18
} 2>&1 | tee gpurun_out/sweep_r04c.txt
