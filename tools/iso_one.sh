#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # the INFV_* knobs below only exist in the experiments build (csrc/knobs.h)
# one isolation run: ./tools/iso_one.sh <INFV_SKIP mask> [extra VAR=value ...]
# (plain timing run: `env VAR=... python` is fine HERE because nothing has touched the GPU before the exec.  Do NOT copy
#  this pattern behind rocprofv3: there the program after `--` must be python3 itself, see tools/pmc_mfma.sh)
m=$1; shift
env INFV_SKIP=$m "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-encode-video > /tmp/iso.json 2>/dev/null
python -c "
import json; d=json.load(open('/tmp/iso.json')); k=d['roofline']['kernel_ms_per_pass']
print('skip', $m, '$*', 'ms', round(d['ms_per_step'],2), {n: round(v,2) for n,v in k.items() if v})"
