#!/bin/bash
# pool_rows2 on scalar-resource buffer loads (56 VGPRs at U=8, 40 at U=4: shares a SIMD with UC waves): wall clock per setting
export INFV_LTM_LIBRARY=exp
tools/env_sweep.sh "INFV_NONE=0" "INFV_PR_U=4" "INFV_PR_PAD=81920" "INFV_PR_U=4 INFV_PR_PAD=81920" "INFV_NONE=1" "INFV_PR_U=4" "INFV_PR_PAD=81920" "INFV_PR_U=4 INFV_PR_PAD=81920" "INFV_PR_PAD=65536" "INFV_POOL_ROWS=0" 2>&1 | tee gpurun_out/sweep_r03w.txt
echo "== residency in situ (U=8, 84K)" | tee -a gpurun_out/sweep_r03w.txt
INFV_WG_STAMPS=1 python tools/residency.py 2>&1 | tail -13 | head -4 | tee -a gpurun_out/sweep_r03w.txt
