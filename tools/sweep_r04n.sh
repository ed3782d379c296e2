#!/bin/bash
# projection GEMM with 90 KB of LDS per workgroup: no pooling workgroup beside it (the pooling keeps full speed, the GEMM its CU)
export INFV_LTM_LIBRARY=exp
for rep in 1 2; do
tools/env_sweep.sh "INFV_NONE=0" "INFV_GEMM_LW_LDS=92160" "INFV_GEMM_LW_LDS=92160 INFV_PR_U=4" "INFV_GEMM_LW_LDS=98304"
done 2>&1 | tee gpurun_out/sweep_r04n.txt
INFV_GEMM_LW_LDS=92160 INFV_WG_STAMPS=1 python tools/residency.py lds90 2>&1 | grep -v amdgpu.ids | tail -18 | tee -a gpurun_out/sweep_r04n.txt
