#!/bin/bash
# wave priorities: pooling above the GEMM's matrix waves, UC above the pooling
export INFV_LTM_LIBRARY=exp INFV_PR_U=4
{
tools/env_sweep.sh "INFV_NONE=0" "INFV_POOL_PRIO=1" "INFV_POOL_PRIO=1 INFV_UC_PRIO=2" "INFV_POOL_PRIO=2 INFV_UC_PRIO=3" "INFV_NONE=1" "INFV_POOL_PRIO=1" "INFV_POOL_PRIO=1 INFV_UC_PRIO=2" "INFV_POOL_PRIO=3 INFV_UC_PRIO=3"
INFV_WG_STAMPS=1 INFV_POOL_PRIO=1 INFV_UC_PRIO=2 python tools/residency.py prio12 2>&1 | grep -v amdgpu.ids | tail -18
} 2>&1 | tee gpurun_out/sweep_r04b.txt
