#!/bin/bash
# with the bf16x6 projection GEMM: who sits where, and is the pooling's workgroup rate (9 per us) a dispatch limit?
export INFV_LTM_LIBRARY=exp
{
INFV_WG_STAMPS=1 python tools/residency.py x6 2>&1 | grep -v amdgpu.ids | tail -20
python tools/launch_table.py gpurun_out/wg_stamps_x6.npy 20 5
tools/env_sweep.sh "INFV_NONE=0" "INFV_PR_WGS=1344" "INFV_PR_PAD=57344" "INFV_PR_U=4" "INFV_PR_U=4 INFV_PR_PAD=57344" "INFV_NONE=1" "INFV_PR_WGS=1344 INFV_PR_PAD=57344"
} 2>&1 | tee gpurun_out/sweep_r05b.txt
