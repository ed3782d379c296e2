#!/bin/bash
# same-box A/B of the pipeline forms (2048-chunk passes, ms): round-4 library, then the new code's variants
one() { local label=$1; shift; echo -n "$label : "
  env "$@" timeout 300 python tools/one_pass.py ${CHUNKS:-2048} 6 2>&1 | grep "^pass" | tail -4 | awk '{print $3}' | sort -n | tr '\n' ' '; echo; }
for r in 1 2; do
one "r04 library                                  " INFV_LTM_LIBRARY=$PWD/infinite-video_amd/libinfv_ltm_r04exp.so
one "per-sub-batch S + pool, atomics (r4 form)    " INFV_LTM_LIBRARY=exp INFV_CHAIN_CALL=0 INFV_CHAIN_XCD=0
one "call-long S (atomics), per-sub-batch pool    " INFV_LTM_LIBRARY=exp INFV_CHAIN_XCD=0 INFV_POOL_CALL=0
one "call-long S (atomics) + call-long pool       " INFV_LTM_LIBRARY=exp INFV_CHAIN_XCD=0
done
