#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}
# shader clock seen by role S (s_memtime cycles over s_memrealtime) in situ, alone, and with subsets of the other kernels
for m in 0 7 6 5 3; do
  echo "== INFV_SKIP=$m"; INFV_SKIP=$m INFV_CHAIN_STAMPS=1 python tools/one_pass.py 2048 4 2>&1 | grep -E "batch-S clock|pass 3" | tail -4
done
