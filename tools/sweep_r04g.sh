#!/bin/bash
# does a smaller padding LDS let a pooling workgroup join a CU where a GEMM workgroup arrived first?
export INFV_LTM_LIBRARY=exp INFV_PR_U=4 INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2
{
for pad in 86016 83968 82944 82000; do
echo "== pad $pad"; INFV_PR_PAD=$pad INFV_WG_STAMPS=1 python tools/residency.py pad$pad 2>&1 | grep -E "span|pool |gemm  |    gemm"
done
tools/env_sweep.sh "INFV_PR_PAD=86016" "INFV_PR_PAD=83968" "INFV_PR_PAD=82944" "INFV_PR_PAD=82000" "INFV_PR_PAD=86016" "INFV_PR_PAD=82944"
} 2>&1 | tee gpurun_out/sweep_r04g.txt
