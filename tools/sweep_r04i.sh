#!/bin/bash
# which other stream stops pooling workgroups from joining CUs that hold a GEMM workgroup?  (priorities on, U=4)
export INFV_LTM_LIBRARY=exp INFV_PR_U=4 INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2 INFV_WG_STAMPS=1
{
for sk in 12 8 4; do
echo "== INFV_SKIP=$sk (1 pool, 2 GEMM, 4 UC+alpha, 8 chain)"; INFV_SKIP=$sk python tools/residency.py skip$sk 2>&1 | grep -v amdgpu.ids | tail -16
done
} 2>&1 | tee gpurun_out/sweep_r04i.txt
