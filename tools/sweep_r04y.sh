#!/bin/bash
# pooling kernel with LDS-DMA loads (INFV_POOL_DMA=1): bit-identity, lifetime beside a GEMM workgroup, wall clock in situ
export INFV_LTM_LIBRARY=exp
{
python -m pytest tests/test_timed_path_gpu.py -x -q -k kept_variants > gpurun_out/variants_r04y.log 2>&1; grep -E "passed|failed|Error|error" gpurun_out/variants_r04y.log | tail -3
echo "== pooling + GEMM only, register loads"; INFV_WG_STAMPS=1 INFV_SKIP=12 python tools/residency.py a 2>&1 | grep -E "pool |gemm  "
echo "== pooling + GEMM only, LDS-DMA loads"; INFV_POOL_DMA=1 INFV_WG_STAMPS=1 INFV_SKIP=12 python tools/residency.py b 2>&1 | grep -E "pool |gemm  "
echo "== alone, LDS-DMA loads"; INFV_POOL_DMA=1 INFV_WG_STAMPS=1 INFV_SKIP=14 python tools/residency.py c 2>&1 | grep -E "pool "
for rep in 1 2; do tools/env_sweep.sh "INFV_NONE=0" "INFV_POOL_DMA=1"; done
} 2>&1 | tee gpurun_out/sweep_r04y.txt
