#!/bin/bash
# pooling kernel as a fixed number of long-lived grid-stride workgroups (no re-dispatch while other kernels' backlogs hold the dispatcher)
export INFV_LTM_LIBRARY=exp
{
tools/env_sweep.sh "INFV_NONE=0" "INFV_PR_WGS=128" "INFV_PR_WGS=160" "INFV_PR_WGS=192" "INFV_PR_WGS=208" "INFV_PR_WGS=256" "INFV_PR_WGS=160 INFV_PR_U=4" "INFV_PR_WGS=192 INFV_PR_U=4" "INFV_PR_WGS=224 INFV_PR_U=4" "INFV_NONE=1"
} 2>&1 | tee gpurun_out/sweep_r04f.txt
