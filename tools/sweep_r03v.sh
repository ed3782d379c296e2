#!/bin/bash
# does the pooling workgroup share a CU with a UC workgroup?  register footprint 3 x 64 (U=8) against 3 x 56 (U=2) per SIMD
export INFV_LTM_LIBRARY=exp INFV_WG_STAMPS=1
{
echo "== pooling + UC, U=8"; INFV_SKIP=10 python tools/residency.py 2>&1 | tail -13 | head -4
echo "== pooling + UC, U=2"; INFV_PR_U=2 INFV_SKIP=10 python tools/residency.py 2>&1 | tail -13 | head -4
echo "== pooling + UC, U=2, pad 80K"; INFV_PR_PAD=81920 INFV_PR_U=2 INFV_SKIP=10 python tools/residency.py 2>&1 | tail -13 | head -4
echo "== pooling + UC, U=8, pad 80K"; INFV_PR_PAD=81920 INFV_SKIP=10 python tools/residency.py 2>&1 | tail -13 | head -4
echo "== in situ U=2"; INFV_PR_U=2 python tools/residency.py 2>&1 | tail -13 | head -4
echo "== in situ U=2, pad 80K"; INFV_PR_PAD=81920 INFV_PR_U=2 python tools/residency.py 2>&1 | tail -13 | head -4
} | tee gpurun_out/sweep_r03v.txt
