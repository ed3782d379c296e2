#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # the INFV_* knobs below only exist in the experiments build (csrc/knobs.h)
# usage (GPU box): tools/skip_probe.sh  -- headline pass with subsets of the streams launched (INFV_SKIP bit mask:
# 1 pooling, 2 projection GEMMs, 4 UC + alpha, 8 role S; results are garbage, timing only): who slows whom
for mask in 0 4 6 2 8 12 14; do
  INFV_SKIP=$mask python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-encode-video --no-selfcheck --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_pass']
print('skip=$mask', 'wall', round(d['ms_per_step'],2), 'pool', k['pool'], 'project', k['project'], 'chain', k['chain'], 'uc', k['uc'])"
done
