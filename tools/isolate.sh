#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # the INFV_* knobs below only exist in the experiments build (csrc/knobs.h)
# which stream slows which: the headline bench with subsets of the kernels launched (INFV_SKIP bit mask:
# 1 = pooling, 2 = projection GEMM, 4 = UC, 8 = role S).  Results are garbage with any bit set; only the times count.
for m in 0 1 2 3 4 5 6 7 8 9 10 12 14 13 11; do
  INFV_SKIP=$m python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-encode-video > /tmp/iso.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/iso.json')); k=d['roofline']['kernel_ms_per_pass']
print('skip', $m, 'ms', round(d['ms_per_step'],2), {n: round(v,2) for n,v in k.items() if v})"
done
