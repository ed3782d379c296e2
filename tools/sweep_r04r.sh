#!/bin/bash
export INFV_LTM_LIBRARY=exp
{
INFV_CU_MASK=64 INFV_PR_PAD=57344 INFV_WG_STAMPS=1 python tools/residency.py mask64 2>&1 | grep -v amdgpu.ids | tail -18
python tools/launch_table.py gpurun_out/wg_stamps_mask64.npy 20 8
} 2>&1 | tee gpurun_out/sweep_r04r.txt
