#!/bin/bash
# same-box alternating A/B of the shippable settings: burst length of the pooling kernel x column slices of the GEMM
export INFV_LTM_LIBRARY=exp
for rep in 1 2 3; do
tools/env_sweep.sh "INFV_PR_U=8" "INFV_PR_U=4" "INFV_PR_U=8 INFV_GEMM_SLICES=2" "INFV_PR_U=4 INFV_GEMM_SLICES=2" "INFV_POOL_ROWS=0"
done 2>&1 | tee gpurun_out/sweep_r04k.txt
python - <<'PY'
import re,collections
d=collections.defaultdict(list)
for l in open("gpurun_out/sweep_r04k.txt"):
    m=re.match(r"sweep \[(.*)\] (\d+) chunks",l)
    if m: d[m.group(1)].append(int(m.group(2)))
for k,v in d.items(): print(k, v, "mean", sum(v)//len(v))
PY
