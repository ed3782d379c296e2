#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # experiment knobs passed in by the caller only exist in the experiments build (csrc/knobs.h)
# usage (GPU box): tools/env_sweep.sh "VAR=a VAR2=b" "VAR=c" ...   -- short headline bench under each environment
for envs in "$@"; do
  out=$(env $envs python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-encode-video --no-selfcheck --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_pass']
print(round(d['value']), 'chunks/s wall', round(d['ms_per_step'],2), 'pool', k['pool'], 'project', k['project'], 'chain', k['chain'], 'uc', k['uc'])")
  echo "sweep [$envs] $out"
done
