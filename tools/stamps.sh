#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # the INFV_* knobs below only exist in the experiments build (csrc/knobs.h)
# in-kernel phase stamps of chain_batch2_kernel (workgroup 0, step 5 of each launch): in situ and alone
echo "== in situ"; INFV_CHAIN_STAMPS=1 python tools/one_pass.py 2048 3 2>&1 | grep -E "batch-S stamps|pass 2" | tail -6
echo "== chain alone (no pooling, GEMM, UC launches)"; INFV_SKIP=7 INFV_CHAIN_STAMPS=1 python tools/one_pass.py 2048 3 2>&1 | grep -E "batch-S stamps|pass 2" | tail -4
echo "== chain + pool"; INFV_SKIP=6 INFV_CHAIN_STAMPS=1 python tools/one_pass.py 2048 3 2>&1 | grep -E "batch-S stamps|pass 2" | tail -4
echo "== RPW=1 in situ"; INFV_CHAIN_RPW=1 INFV_CHAIN_STAMPS=1 python tools/one_pass.py 2048 3 2>&1 | grep -E "batch-S stamps|pass 2" | tail -4
echo "== RPW=1 alone"; INFV_CHAIN_RPW=1 INFV_SKIP=7 INFV_CHAIN_STAMPS=1 python tools/one_pass.py 2048 3 2>&1 | grep -E "batch-S stamps|pass 2" | tail -4
