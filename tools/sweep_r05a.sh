#!/bin/bash
# projection GEMM as six bf16 MFMA products of exact three-piece splits (default) against the fp32-MFMA GEMM (INFV_PROJ_FP32=1)
{
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_r05a.log 2>&1; grep -E "passed|failed" gpurun_out/gpu_tests_r05a.log | tail -2
for rep in 1 2 3; do
tools/env_sweep.sh "INFV_PROJ_FP32=1" "INFV_PROJ_FP32=0"
done
INFV_LTM_LIBRARY=exp INFV_WG_STAMPS=1 python tools/residency.py x6 2>&1 | grep -v amdgpu.ids | tail -18
python tools/launch_table.py gpurun_out/wg_stamps_x6.npy 20 6
} 2>&1 | tee gpurun_out/sweep_r05a.txt
