#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # experiment knobs passed in by the caller only exist in the experiments build (csrc/knobs.h)
# usage (GPU box): tools/short_sweep.sh "<ENV=V ...>" ...   -- ms per 256-chunk call (the 8-GPU shard) under each env set
for envs in "$@"; do
  ( for kv in $envs; do export "$kv"; done
    python3 tools/one_pass.py 256 12 | tail -8 | awk '{print $3}' | sort -n | awk -v e="$envs" '{s[NR]=$1} END{print e, "median ms", s[int((NR+1)/2)], "min", s[1]}' )
done
