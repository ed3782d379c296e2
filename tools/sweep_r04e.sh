#!/bin/bash
# HIP runtime hardware-queue count: do the pipeline's four streams share AQL queues / CP pipes?
export INFV_LTM_LIBRARY=exp
{
tools/env_sweep.sh "INFV_NONE=0" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=2" "GPU_MAX_HW_QUEUES=16" "INFV_NONE=1" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=6" "GPU_MAX_HW_QUEUES=8 INFV_PR_U=4"
} 2>&1 | tee gpurun_out/sweep_r04e.txt
