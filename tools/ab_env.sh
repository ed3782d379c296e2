#!/bin/bash
# Same-box A/B of environment settings on one library: alternating runs of tools/one_pass.py, sorted pass times (ms) per run.
# usage: tools/ab_env.sh <rounds> <chunks> "ENV=V ..." "ENV=V ..." ...     (INFV_LTM_LIBRARY defaults to exp)
rounds=$1; chunks=$2; shift 2
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}
for r in $(seq 1 $rounds); do
  for envs in "$@"; do
    ( for kv in $envs; do export "$kv"; done
      echo -n "$envs : "
      timeout 600 python tools/one_pass.py $chunks 8 2>&1 | grep "^pass" | tail -6 | awk '{print $3}' | sort -n | tr '\n' ' '; echo )
  done
done
