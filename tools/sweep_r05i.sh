#!/bin/bash
# sub-batch size of long calls, shipped library, same box alternating
run() {
  out=$(python bench.py --steps 8 --warmup 2 --batch-chunks $1 --no-cpu-baseline --no-encode-video --no-selfcheck --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_pass']
print(round(d['value']), 'chunks/s wall', round(d['ms_per_step'],2), 'pool', k['pool'], 'project', k['project'], 'chain', k['chain'], 'uc', k['uc'])")
  echo "sweep [batch $1] $out"
}
for rep in 1 2 3; do run 42; run 64; run 84; done 2>&1 | tee gpurun_out/sweep_r05i.txt
