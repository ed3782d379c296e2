#!/bin/bash
# padding of 56 KB: a GEMM (72 KB), a pooling (56 KB) and an alpha (29 KB) workgroup fit one CU together; with / without wave priorities
export INFV_LTM_LIBRARY=exp
for rep in 1 2; do
tools/env_sweep.sh "INFV_NONE=0" "INFV_PR_PAD=57344" "INFV_PR_PAD=57344 INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2" "INFV_PR_PAD=57344 INFV_POOL_PRIO=1 INFV_UC_PRIO=1 INFV_ALPHA_PRIO=1" "INFV_PR_PAD=57344 INFV_PR_U=4 INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2"
done 2>&1 | tee gpurun_out/sweep_r04o.txt
INFV_PR_PAD=57344 INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2 INFV_WG_STAMPS=1 python tools/residency.py pad56prio 2>&1 | grep -v amdgpu.ids | tail -18 | tee -a gpurun_out/sweep_r04o.txt
