#!/bin/bash
# residency of the pooling workgroups: in situ, alone, and beside each of the other three streams
export INFV_LTM_LIBRARY=exp INFV_WG_STAMPS=1
{
echo "== in situ"; RES_TAG=insitu python tools/residency.py 2>&1 | tail -14
echo "== pooling alone (INFV_SKIP=14: everything else skipped)"; RES_TAG=alone INFV_SKIP=14 python tools/residency.py 2>&1 | tail -14
echo "== pooling + role S (INFV_SKIP=6)"; RES_TAG=chain INFV_SKIP=6 python tools/residency.py 2>&1 | tail -14
echo "== pooling + GEMM (INFV_SKIP=12)"; RES_TAG=gemm INFV_SKIP=12 python tools/residency.py 2>&1 | tail -14
echo "== pooling + UC (INFV_SKIP=10)"; RES_TAG=uc INFV_SKIP=10 python tools/residency.py 2>&1 | tail -14
} | tee gpurun_out/sweep_r03u.txt
