#!/bin/bash
# Role-S A/B on the GPU box (experiments build): in-kernel phase stamps of the persistent chain kernel (workgroup 0, step 5 of
# each launch) and the pass time, alone (no pooling / GEMM / UC launches) and in situ, per (rows per wave, exchange shards).
# usage: tools/chain_ab.sh "<rpw>:<shards> ..." [chunks]
export INFV_LTM_LIBRARY=exp
cfgs=${1:-"2:1"}; chunks=${2:-2048}
for c in $cfgs; do
  rpw=${c%%:*}; sh=${c##*:}
  echo "== rpw $rpw shards $sh: alone"
  INFV_CHAIN_RPW=$rpw INFV_CHAIN_SHARDS=$sh INFV_SKIP=7 INFV_CHAIN_STAMPS=1 python tools/one_pass.py $chunks 3 2>&1 | grep -E "batch-S stamps|batch-S avg|pass 2|rror" | tail -7
  echo "== rpw $rpw shards $sh: in situ"
  INFV_CHAIN_RPW=$rpw INFV_CHAIN_SHARDS=$sh INFV_CHAIN_STAMPS=1 python tools/one_pass.py $chunks 4 2>&1 | grep -E "batch-S stamps|batch-S avg|pass [23]|rror" | tail -9
done
