#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # the INFV_* knobs below only exist in the experiments build (csrc/knobs.h)
for fl in "$@"; do
  INFV_S_FLAGS=$fl INFV_SKIP=6 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-encode-video --no-selfcheck --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_pass']
print('flags $fl  S+pool only: wall', round(d['ms_per_step'],2), 'pool', k['pool'], 'chain', k['chain'])"
done
