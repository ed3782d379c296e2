#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # the INFV_* knobs below only exist in the experiments build (csrc/knobs.h)
tools/env_sweep.sh \
 "INFV_POOL_ROWS=0 INFV_CHAIN_RPW=1" \
 "INFV_POOL_ROWS=0" \
 "INFV_POOL_ROWS=0 INFV_VPROJ_ON_UC=0" \
 "INFV_PR_NT=256 INFV_PR_U=8" \
 "INFV_PR_NT=256 INFV_PR_U=8 INFV_VPROJ_ON_UC=0" \
 "INFV_PR_NT=512 INFV_PR_U=4" \
 "INFV_PR_NT=512 INFV_PR_U=8" \
 "INFV_POOL_ROWS=0 INFV_WHOLE_CALL=1" \
 "INFV_PR_NT=256 INFV_PR_U=8 INFV_WHOLE_CALL=1" \
 2>&1 | tee gpurun_out/sweep_r03e.txt
