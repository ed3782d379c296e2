#!/bin/bash
# CU masks: the first K CUs reserved for role S (its own masked stream), the worker streams masked to the rest
export INFV_LTM_LIBRARY=exp
{
INFV_CU_MASK=56 INFV_WG_STAMPS=1 timeout 300 python tools/residency.py mask56 2>&1 | grep -v amdgpu.ids | tail -18
python - <<'PY'
import numpy as np
st=np.load("gpurun_out/wg_stamps_mask56.npy"); st=st[(st[:,1]>0)&(st[:,0]>0)]
hw=st[:,2]&0xffffffff; xcc=st[:,2]>>32; cu=(xcc<<8)|(((hw>>13)&7)<<5)|((hw>>8)&15); k=st[:,3]
rs=set(cu[k==4]); others=set(cu[(k!=4)])
print("CUs used by role S:", len(rs), " by the other kernels:", len(others), " shared:", len(rs&others))
PY
tools/env_sweep.sh "INFV_NONE=0" "INFV_CU_MASK=48" "INFV_CU_MASK=56" "INFV_CU_MASK=56 INFV_PR_PAD=57344" "INFV_CU_MASK=56 INFV_PR_PAD=40960" "INFV_CU_MASK=64 INFV_PR_PAD=57344" "INFV_NONE=1"
} 2>&1 | tee gpurun_out/sweep_r04p.txt
