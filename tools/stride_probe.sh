#!/bin/bash
# usage (GPU box): tools/stride_probe.sh <stride> ...  -- rebuilds the library with INFV_ACC_STRIDE and runs the short bench
for st in "$@"; do
  INFV_CXXFLAGS="-DINFV_ACC_STRIDE=$st" python -c "import __graft_entry__ as g; g.build(force=True)" > /dev/null 2>&1
  tools/quick_bench.sh stride$st 8
  INFV_SKIP=6 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-encode-video --no-selfcheck --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_pass']
print('   stride $st  S+pool only: wall', round(d['ms_per_step'],2), 'pool', k['pool'], 'chain', k['chain'])"
done
