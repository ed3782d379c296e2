#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # experiment knobs passed in by the caller only exist in the experiments build (csrc/knobs.h)
# sub-batch size sweep of the headline bench: ./tools/sweep_batch.sh <chunks> <batch sizes...>
chunks=$1; shift
for b in "$@"; do
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-encode-video --chunks $chunks --batch-chunks $b > /tmp/sweep.json
  python -c "import json; d=json.load(open('/tmp/sweep.json')); print('chunks', $chunks, 'batch', $b, 'chunks/s', round(d['value']), 'ms', round(d['ms_per_step'], 3))"
done
