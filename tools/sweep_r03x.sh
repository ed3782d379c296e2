#!/bin/bash
# residency of the pooling workgroups with the 40-VGPR instantiation (U=4), in situ and beside UC / GEMM only
export INFV_LTM_LIBRARY=exp INFV_WG_STAMPS=1 INFV_PR_U=4
{
echo "== in situ U=4"; python tools/residency.py 2>&1 | tail -13
echo "== pooling + UC U=4"; INFV_SKIP=10 python tools/residency.py 2>&1 | tail -13 | head -4
echo "== pooling + GEMM U=4"; INFV_SKIP=12 python tools/residency.py 2>&1 | tail -13 | head -4
echo "== pooling + UC + GEMM U=4 (no chain)"; INFV_SKIP=8 python tools/residency.py 2>&1 | tail -13 | head -4
echo "== alone U=4"; INFV_SKIP=14 python tools/residency.py 2>&1 | tail -13 | head -4
echo "== alone U=8"; INFV_PR_U=8 INFV_SKIP=14 python tools/residency.py 2>&1 | tail -13 | head -4
} | tee gpurun_out/sweep_r03x.txt
