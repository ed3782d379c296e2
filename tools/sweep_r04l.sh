#!/bin/bash
# (after removing the GEMM's experiment branches, which had cost the experiments build's GEMM 202 VGPRs against the shipped 130
#  and with them its seat beside a pooling workgroup: sweeps r03y..r04k ran with that handicap)
# burst length, wave priorities and GEMM column slices again, same box, alternating
export INFV_LTM_LIBRARY=exp
for rep in 1 2; do
tools/env_sweep.sh "INFV_PR_U=8" "INFV_PR_U=4" "INFV_PR_U=8 INFV_POOL_PRIO=1" "INFV_PR_U=4 INFV_POOL_PRIO=1" "INFV_PR_U=4 INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2" "INFV_PR_U=8 INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2" "INFV_PR_U=8 INFV_GEMM_SLICES=2" "INFV_POOL_ROWS=0"
done 2>&1 | tee gpurun_out/sweep_r04l.txt
INFV_LTM_LIBRARY= python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-encode-video --no-selfcheck --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('shipped library', round(d['value']), 'chunks/s', d['roofline']['kernel_ms_per_pass'])" | tee -a gpurun_out/sweep_r04l.txt
python - <<'PY' | tee -a gpurun_out/sweep_r04l.txt
import re,collections
d=collections.defaultdict(list)
for l in open("gpurun_out/sweep_r04l.txt"):
    m=re.match(r"sweep \[(.*)\] (\d+) chunks",l)
    if m: d[m.group(1)].append(int(m.group(2)))
for k,v in d.items(): print(k, v, "mean", sum(v)//len(v))
PY
for cfg in "INFV_PR_U=8" "INFV_PR_U=8 INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2"; do
echo "== residency [$cfg]" | tee -a gpurun_out/sweep_r04l.txt
env $cfg INFV_WG_STAMPS=1 python tools/residency.py l 2>&1 | grep -v amdgpu.ids | tail -18 | tee -a gpurun_out/sweep_r04l.txt
done
