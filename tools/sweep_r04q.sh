#!/bin/bash
# five rotating workspace sets (projection GEMM up to five sub-batches ahead of the UC kernel) with and without CU masks
export INFV_LTM_LIBRARY=exp
{
python -m pytest tests/test_timed_path_gpu.py -x -q -k "odd_call or oracle or equals" 2>&1 | tail -2
tools/env_sweep.sh "INFV_NONE=0" "INFV_CU_MASK=64" "INFV_CU_MASK=64 INFV_PR_PAD=57344" "INFV_CU_MASK=64 INFV_PR_PAD=40960" "INFV_CU_MASK=60 INFV_PR_PAD=57344" "INFV_CU_MASK=72 INFV_PR_PAD=57344" "INFV_PR_PAD=57344" "INFV_NONE=1"
} 2>&1 | tee gpurun_out/sweep_r04q.txt
