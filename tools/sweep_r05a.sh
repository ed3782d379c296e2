#!/bin/bash
# (record of how sweep_r05a/d.txt were produced: at that time the bf16x6 GEMM was the default, INFV_PROJ_FP32=1 selected the fp32-MFMA GEMM and
#  INFV_X6_PIPE chose between the two bf16x6 kernels; now INFV_PROJ_X6=1 opts in and only the single-tile kernel is kept)
# projection GEMM as six bf16 MFMA products of exact three-piece splits (default) against the fp32-MFMA GEMM (INFV_PROJ_FP32=1)
{
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_r05a.log 2>&1; grep -E "passed|failed" gpurun_out/gpu_tests_r05a.log | tail -2
for rep in 1 2 3; do
tools/env_sweep.sh "INFV_PROJ_FP32=1" "INFV_PROJ_FP32=0"
done
INFV_LTM_LIBRARY=exp INFV_WG_STAMPS=1 python tools/residency.py x6 2>&1 | grep -v amdgpu.ids | tail -18
python tools/launch_table.py gpurun_out/wg_stamps_x6.npy 20 6
} 2>&1 | tee gpurun_out/sweep_r05a.txt
