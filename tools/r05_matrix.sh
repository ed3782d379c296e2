#!/bin/bash
# same-box A/B (2048-chunk passes, ms): round-4 library, the shipped default, and round 5's call-long forms (experiments build)
one() { local label=$1; shift; echo -n "$label : "
  env "$@" timeout 300 python tools/one_pass.py ${CHUNKS:-2048} 6 2>&1 | grep "^pass" | tail -4 | awk '{print $3}' | sort -n | tr '\n' ' '; echo; }
for r in 1 2; do
one "r04 library                                        " INFV_LTM_LIBRARY=$PWD/infinite-video_amd/libinfv_ltm_r04exp.so
one "shipped default (planes by the pooling kernel)     " INFV_LTM_LIBRARY=
one "  ... with split3_rows_kernel (INFV_POOL_PLANES=0) " INFV_LTM_LIBRARY=exp INFV_POOL_PLANES=0
one "call-long S                                        " INFV_LTM_LIBRARY=exp INFV_CHAIN_CALL=1
one "call-long S + pool                                 " INFV_LTM_LIBRARY=exp INFV_CHAIN_CALL=1 INFV_POOL_CALL=1
one "call-long S + pool + GEMM (32)                     " INFV_LTM_LIBRARY=exp INFV_CHAIN_CALL=1 INFV_POOL_CALL=1 INFV_GEMM_CALL=1
done
