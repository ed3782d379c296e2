#!/bin/bash
# who shares a CU with whom: residency of every pipeline kernel's workgroups in one headline pass
export INFV_LTM_LIBRARY=exp INFV_WG_STAMPS=1
{
python tools/residency.py insitu_u8 2>&1 | grep -v amdgpu.ids | tail -18
INFV_PR_U=4 python tools/residency.py insitu_u4 2>&1 | grep -v amdgpu.ids | tail -18
} | tee gpurun_out/sweep_r04a.txt
