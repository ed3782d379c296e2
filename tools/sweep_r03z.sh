#!/bin/bash
# s_setprio(3) in the pooling kernel: lifetime of its workgroups beside the GEMM, then wall clock in situ
export INFV_LTM_LIBRARY=exp
{
for pr in 0 1; do echo "== pooling + GEMM, U=4, INFV_POOL_PRIO=$pr"; INFV_WG_STAMPS=1 INFV_PR_U=4 INFV_SKIP=12 INFV_POOL_PRIO=$pr python tools/residency.py 2>&1 | tail -13 | head -4; done
for pr in 0 1; do echo "== in situ, U=4, INFV_POOL_PRIO=$pr"; INFV_WG_STAMPS=1 INFV_PR_U=4 INFV_POOL_PRIO=$pr python tools/residency.py 2>&1 | tail -13 | head -4; done
tools/env_sweep.sh "INFV_POOL_PRIO=0 INFV_PR_U=4" "INFV_POOL_PRIO=1 INFV_PR_U=4" "INFV_POOL_PRIO=0 INFV_PR_U=8" "INFV_POOL_PRIO=1 INFV_PR_U=8" "INFV_POOL_PRIO=0 INFV_PR_U=4" "INFV_POOL_PRIO=1 INFV_PR_U=4" "INFV_POOL_PRIO=1 INFV_PR_U=4 INFV_PR_PAD=81920" "INFV_POOL_PRIO=1 INFV_PR_U=8 INFV_PR_PAD=81920"
} 2>&1 | tee gpurun_out/sweep_r03z.txt
