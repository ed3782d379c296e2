#!/bin/bash
one() { local label=$1; shift; echo -n "$label : "
  env "$@" timeout 300 python tools/one_pass.py ${CHUNKS:-2048} 6 2>&1 | grep "^pass" | tail -4 | awk '{print $3}' | sort -n | tr '\n' ' '; echo; }
B="INFV_LTM_LIBRARY=exp INFV_CHAIN_XCD=0 INFV_CHAIN_CALL=0"
for r in 1 2; do
one "r4 form                      " $B
one "r4 form, no alpha            " $B INFV_SKIP=16
one "r4 form, no UC               " $B INFV_SKIP=32
one "r4 form, no alpha, no UC     " $B INFV_SKIP=4
one "r4 form, no GEMM             " $B INFV_SKIP=2
done
