#!/bin/bash
# wave priorities: everything above the GEMM's matrix waves (role S stays at 3)
export INFV_LTM_LIBRARY=exp INFV_PR_U=4
{
tools/env_sweep.sh "INFV_NONE=0" "INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2" "INFV_POOL_PRIO=1 INFV_UC_PRIO=1 INFV_ALPHA_PRIO=1" "INFV_ALPHA_PRIO=1" "INFV_NONE=1" "INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2" "INFV_POOL_PRIO=1 INFV_UC_PRIO=1 INFV_ALPHA_PRIO=1" "INFV_ALPHA_PRIO=1"
INFV_WG_STAMPS=1 INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2 python tools/residency.py prio122 2>&1 | grep -v amdgpu.ids | tail -18
} 2>&1 | tee gpurun_out/sweep_r04d.txt
