#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # the INFV_* knobs below only exist in the experiments build (csrc/knobs.h)
tools/env_sweep.sh "INFV_NONE=0" "INFV_POOL_DB=1" "INFV_POOL_DB=2" "INFV_POOL_DB=4" "INFV_NONE=1" "INFV_POOL_DB=2 INFV_POOL_PAD=100000" "INFV_POOL_DB=1 INFV_POOL_NT=512" 2>&1 | tee gpurun_out/sweep_r03l.txt
python -m pytest tests/test_ltm_gpu.py -x -q -k "pool or pieces or pooled" 2>&1 | tail -2
