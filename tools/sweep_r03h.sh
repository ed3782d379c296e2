#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # the INFV_* knobs below only exist in the experiments build (csrc/knobs.h)
export INFV_VPROJ_ON_UC=0 INFV_POOL_ROWS=0
tools/env_sweep.sh \
 "INFV_POOL_PAD=86016" \
 "INFV_POOL_PAD=76000" \
 "INFV_POOL_PAD=64000" \
 "INFV_POOL_PAD=56000" \
 "INFV_POOL_PAD=48000" \
 "INFV_POOL_PAD=56000 INFV_POOL_UNROLL=2" \
 "INFV_POOL_PAD=40000 INFV_POOL_UNROLL=2" \
 "INFV_POOL_PAD=56000 INFV_POOL_UNROLL=8" \
 "INFV_POOL_PAD=100000" \
 2>&1 | tee gpurun_out/sweep_r03h.txt
