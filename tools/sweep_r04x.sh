#!/bin/bash
# what of a pooling workgroup suffers beside a GEMM workgroup: its adds or its loads?  pooling + GEMM only; the second library's
# pooling kernel keeps its loads and drops its adds (-DINFV_POOL_NOADD, timing only)
export INFV_WG_STAMPS=1 INFV_SKIP=12
{
echo "== default pooling kernel"; INFV_LTM_LIBRARY=exp python tools/residency.py add 2>&1 | grep -E "pool |gemm  |    gemm|    pool"
echo "== no adds"; INFV_LTM_LIBRARY=$PWD/tools/ab/lib_noadd.so python tools/residency.py noadd 2>&1 | grep -E "pool |gemm  |    gemm|    pool"
echo "== no adds, alone (INFV_SKIP=14)"; INFV_SKIP=14 INFV_LTM_LIBRARY=$PWD/tools/ab/lib_noadd.so python tools/residency.py noadd_alone 2>&1 | grep -E "pool "
echo "== default, alone (INFV_SKIP=14)"; INFV_SKIP=14 INFV_LTM_LIBRARY=exp python tools/residency.py add_alone 2>&1 | grep -E "pool "
} 2>&1 | tee gpurun_out/sweep_r04x.txt
