#!/bin/bash
# (record: INFV_X6_PIPE selected a pipelined bf16x6 kernel that was not kept)
# bf16x6 projection GEMM as a short exclusive burst: the pipelined kernel with six tiles in flight (244 registers, 72 KB: two
# workgroups per CU, no room for a pooling workgroup beside them)
export INFV_LTM_LIBRARY=exp INFV_PROJ_X6=1
{
for rep in 1 2; do
tools/env_sweep.sh "INFV_X6_PIPE=0" "INFV_X6_PIPE=1"
done
INFV_X6_PIPE=1 INFV_WG_STAMPS=1 python tools/residency.py x6b 2>&1 | grep -v amdgpu.ids | tail -20
python tools/launch_table.py gpurun_out/wg_stamps_x6b.npy 20 5
} 2>&1 | tee gpurun_out/sweep_r05e.txt
