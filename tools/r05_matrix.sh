#!/bin/bash
# same-box A/B (2048-chunk passes, ms)
one() { local label=$1; shift; echo -n "$label : "
  env "$@" timeout 300 python tools/one_pass.py ${CHUNKS:-2048} 6 2>&1 | grep "^pass" | tail -4 | awk '{print $3}' | sort -n | tr '\n' ' '; echo; }
B="INFV_LTM_LIBRARY=exp INFV_CHAIN_XCD=0"
for r in 1 2; do
one "r04 library                                   " INFV_LTM_LIBRARY=$PWD/infinite-video_amd/libinfv_ltm_r04exp.so
one "all call-long (S atomics, pool, GEMM 32)      " $B
one " + pool pad 72K, alpha LDS 88K                " $B INFV_PR_PAD=73728 INFV_ALPHA_LDS=90112
one " + pool pad 72K only                          " $B INFV_PR_PAD=73728
one " + alpha LDS 76K (one per CU beside the pool) " $B INFV_ALPHA_LDS=77824
one " + pool pad 72K, alpha 88K, GEMM 36           " $B INFV_PR_PAD=73728 INFV_ALPHA_LDS=90112 INFV_GEMM_WGS=36
done
