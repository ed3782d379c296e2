#!/bin/bash
# headline bench under different sub-batch sizes (engine parameter max_batch_chunks): tools/batch_sweep.sh 42 48 54 ...
for bc in "$@"; do
  out=$(python bench.py --steps 6 --warmup 2 --batch-chunks $bc --no-cpu-baseline --no-encode-video --no-selfcheck --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_pass']
print(round(d['value']), 'chunks/s wall', round(d['ms_per_step'],2), 'pool', k['pool'], 'project', k['project'], 'chain', k['chain'], 'uc', k['uc'], 'shard256', d.get('shard256_ms'))")
  echo "batch-chunks $bc: $out"
done
