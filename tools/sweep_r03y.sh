#!/bin/bash
# what in the projection GEMM stretches a co-resident pooling workgroup?  pooling + GEMM only, GEMM variants (timing only):
# X=1 one MFMA per tile, X=2 loaders move nothing, X=4 MFMA waves sleep instead, X=6 sleep + no loads (LDS reads and barriers only)
export INFV_LTM_LIBRARY=exp INFV_WG_STAMPS=1 INFV_PR_U=4 INFV_SKIP=12
{
for x in 0 1 2 3 4 6; do echo "== pooling + GEMM, INFV_GEMM_X=$x"; INFV_GEMM_X=$x python tools/residency.py 2>&1 | tail -13 | head -4; done
} | tee gpurun_out/sweep_r03y.txt
