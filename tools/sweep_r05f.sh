#!/bin/bash
# more pooling seats (padding 44 / 56 KB, 40-register instantiation) with fewer role-S launches (84- / 126-chunk sub-batches)
export INFV_LTM_LIBRARY=exp
run() {  # $1 = batch chunks, rest = env
  bc=$1; shift
  out=$(env "$@" python bench.py --steps 6 --warmup 2 --batch-chunks $bc --no-cpu-baseline --no-encode-video --no-selfcheck --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_pass']
print(round(d['value']), 'chunks/s wall', round(d['ms_per_step'],2), 'pool', k['pool'], 'project', k['project'], 'chain', k['chain'], 'uc', k['uc'])")
  echo "sweep [batch $bc $*] $out"
}
{
run 42 INFV_NONE=0
run 84 INFV_PR_PAD=57344
run 84 INFV_PR_PAD=45056 INFV_PR_U=4
run 126 INFV_PR_PAD=57344
run 126 INFV_PR_PAD=45056 INFV_PR_U=4
run 168 INFV_PR_PAD=45056 INFV_PR_U=4
run 84 INFV_NONE=0
} 2>&1 | tee gpurun_out/sweep_r05f.txt
