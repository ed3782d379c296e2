#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # the INFV_* knobs below only exist in the experiments build (csrc/knobs.h)
export INFV_VPROJ_ON_UC=0 INFV_POOL_ROWS=0
tools/env_sweep.sh \
 "INFV_NONE=0" \
 "INFV_WHOLE_CALL=1" \
 "INFV_WHOLE_CALL=1 INFV_POOL_UNROLL=8" \
 "INFV_WHOLE_CALL=1 INFV_SUB_BATCH=32" \
 "INFV_WHOLE_CALL=1 INFV_SUB_BATCH=21" \
 "INFV_WHOLE_CALL=1 INFV_POOL_ROWS=1 INFV_PR_NT=512 INFV_PR_U=2" \
 2>&1 | tee gpurun_out/sweep_r03g.txt
