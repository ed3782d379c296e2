#!/bin/bash
# (record of how sweep_r05a/d.txt were produced: at that time the bf16x6 GEMM was the default, INFV_PROJ_FP32=1 selected the fp32-MFMA GEMM and
#  INFV_X6_PIPE chose between the two bf16x6 kernels; now INFV_PROJ_X6=1 opts in and only the single-tile kernel is kept)
# pipelined bf16x6 projection GEMM (default) against the single-tile one (INFV_X6_PIPE=0) and the fp32-MFMA GEMM (INFV_PROJ_FP32=1)
export INFV_LTM_LIBRARY=exp
{
python -m pytest tests/test_ltm_gpu.py -x -q -k "bf16x6" 2>&1 | grep -E "passed|failed" | tail -1
python -m pytest tests/test_ltm_gpu.py tests/test_timed_path_gpu.py -x -q --deselect tests/test_timed_path_gpu.py::test_kept_variants_reproduce_the_default_bit_for_bit > gpurun_out/gpu_tests_r05d.log 2>&1; grep -E "passed|failed" gpurun_out/gpu_tests_r05d.log | tail -1
for rep in 1 2; do
tools/env_sweep.sh "INFV_PROJ_FP32=1" "INFV_X6_PIPE=0" "INFV_NONE=0"
done
INFV_WG_STAMPS=1 python tools/residency.py x6p 2>&1 | grep -v amdgpu.ids | tail -20
python tools/launch_table.py gpurun_out/wg_stamps_x6p.npy 20 5
} 2>&1 | tee gpurun_out/sweep_r05d.txt
