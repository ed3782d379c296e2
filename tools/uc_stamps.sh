#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}
# in-kernel phase times of one V' workgroup of uc_fast_kernel (x10 ns), in situ and alone
echo "== in situ"; INFV_UC_STAMPS=1 python tools/one_pass.py 2048 3 2>&1 | grep -E "uc stamps|pass 2" | tail -4
echo "== UC alone (no pooling, GEMM, chain: garbage inputs, timing only)"; INFV_SKIP=11 INFV_UC_STAMPS=1 python tools/one_pass.py 2048 3 2>&1 | grep -E "uc stamps|pass 2" | tail -3
