#!/bin/bash
one() { local label=$1; shift; echo -n "$label : "
  env "$@" timeout 300 python tools/one_pass.py ${CHUNKS:-2048} 8 2>&1 | grep "^pass" | tail -5 | awk '{print $3}' | sort -n | tr '\n' ' '; echo; }
B="INFV_LTM_LIBRARY=exp INFV_CHAIN_XCD=0"
for r in 1 2; do
CHUNKS=256 one "256: r04 library        " INFV_LTM_LIBRARY=$PWD/infinite-video_amd/libinfv_ltm_r04exp.so
CHUNKS=256 one "256: call-long, GEMM 32 " $B
CHUNKS=256 one "256: call-long, GEMM 48 " $B INFV_GEMM_WGS=48
CHUNKS=256 one "256: call-long, GEMM 56 " $B INFV_GEMM_WGS=56
CHUNKS=256 one "256: call-long, GEMM 64 " $B INFV_GEMM_WGS=64
CHUNKS=256 one "256: call-long, GEMM per batch " $B INFV_GEMM_CALL=0
CHUNKS=256 one "256: call-long S only   " $B INFV_POOL_CALL=0
done
one "2048: call-long, GEMM 48" $B INFV_GEMM_WGS=48
one "2048: call-long, GEMM 36" $B INFV_GEMM_WGS=36
