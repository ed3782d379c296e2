#!/bin/bash
# chain-side potential: the same passes without the pooling stream (INFV_SKIP=1: garbage rows, real timing of everything else)
one() { local label=$1; shift; echo -n "$label : "
  env "$@" timeout 300 python tools/one_pass.py ${CHUNKS:-2048} 6 2>&1 | grep "^pass" | tail -4 | awk '{print $3}' | sort -n | tr '\n' ' '; echo; }
for r in 1 2; do
one "no pool: per-sub-batch, atomics (r4 form)   " INFV_LTM_LIBRARY=exp INFV_SKIP=1 INFV_CHAIN_CALL=0 INFV_CHAIN_XCD=0
one "no pool: call-long, atomics                 " INFV_LTM_LIBRARY=exp INFV_SKIP=1 INFV_CHAIN_XCD=0
one "no pool: call-long, sc1 mailboxes, linear   " INFV_LTM_LIBRARY=exp INFV_SKIP=1 INFV_CHAIN_LINEAR=1
one "no pool: call-long, XCD-local mailboxes     " INFV_LTM_LIBRARY=exp INFV_SKIP=1
one "chain only: per-sub-batch, atomics          " INFV_LTM_LIBRARY=exp INFV_SKIP=7 INFV_CHAIN_CALL=0 INFV_CHAIN_XCD=0
one "chain only: call-long, atomics              " INFV_LTM_LIBRARY=exp INFV_SKIP=7 INFV_CHAIN_XCD=0
one "chain only: call-long, sc1 mailboxes linear " INFV_LTM_LIBRARY=exp INFV_SKIP=7 INFV_CHAIN_LINEAR=1
one "chain only: call-long, XCD-local mailboxes  " INFV_LTM_LIBRARY=exp INFV_SKIP=7
one "pool + chain: per-sub-batch, atomics        " INFV_LTM_LIBRARY=exp INFV_SKIP=6 INFV_CHAIN_CALL=0 INFV_CHAIN_XCD=0
one "pool + chain: call-long, atomics            " INFV_LTM_LIBRARY=exp INFV_SKIP=6 INFV_CHAIN_XCD=0
done
