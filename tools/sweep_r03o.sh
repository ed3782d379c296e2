#!/bin/bash
# sub-batch size with the round-3 structure (fewer role-S launches per video vs longer pipeline fill)
for bc in 42 56 63 84 42 28; do
  echo -n "batch-chunks $bc: "
  python bench.py --steps 8 --warmup 2 --batch-chunks $bc --no-cpu-baseline --no-encode-video --no-selfcheck --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_pass']
print(round(d['value']), 'chunks/s wall', round(d['ms_per_step'],2), 'pool', k['pool'], 'project', k['project'], 'chain', k['chain'], 'uc', k['uc'])"
done 2>&1 | tee gpurun_out/sweep_r03o.txt
