#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # the INFV_* knobs below only exist in the experiments build (csrc/knobs.h)
# round-3 sweep b: the fused pool+rows kernel, its footprint (threads / loads per group / padding LDS) and who may sit beside it
tools/env_sweep.sh \
 "INFV_POOL_ROWS=0" \
 "INFV_NONE=0" \
 "INFV_PR_NT=256 INFV_PR_U=4 INFV_PR_PAD=0" \
 "INFV_PR_NT=256 INFV_PR_U=4 INFV_PR_PAD=0 INFV_S_LDS=150000" \
 "INFV_PR_NT=256 INFV_PR_U=8 INFV_PR_PAD=0" \
 "INFV_PR_NT=256 INFV_PR_U=4 INFV_PR_PAD=13000 INFV_S_LDS=150000" \
 "INFV_PR_NT=256 INFV_PR_U=4 INFV_PR_PAD=0 INFV_S_LDS=150000 INFV_WHOLE_CALL=1" \
 "INFV_PR_NT=256 INFV_PR_U=4 INFV_PR_PAD=40000" \
 "INFV_PR_NT=256 INFV_PR_U=4 INFV_PR_PAD=40000 INFV_S_LDS=150000" \
 "INFV_PR_NT=256 INFV_PR_U=8 INFV_PR_PAD=40000" \
 "INFV_PR_NT=512 INFV_PR_U=4 INFV_PR_PAD=60000" \
 "INFV_PR_NT=512 INFV_PR_U=8" \
 "INFV_PR_NT=256 INFV_PR_U=2 INFV_PR_PAD=0 INFV_S_LDS=150000" \
 "INFV_PR_NT=256 INFV_PR_U=8 INFV_PR_PAD=84000" \
 "INFV_PR_NT=256 INFV_PR_U=8 INFV_PR_PAD=60000" \
 2>&1 | tee gpurun_out/sweep_r03b.txt
