#!/bin/bash
# fewer, longer pooling workgroups: 2 / 3 / 4 rows per workgroup (grid-stride), default padding
export INFV_LTM_LIBRARY=exp
for rep in 1 2; do
tools/env_sweep.sh "INFV_NONE=0" "INFV_PR_WGS=1344" "INFV_PR_WGS=896" "INFV_PR_WGS=672"
done 2>&1 | tee gpurun_out/sweep_r04u.txt
