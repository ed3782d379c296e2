#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # the INFV_* knobs below only exist in the experiments build (csrc/knobs.h)
python -m pytest tests/test_timed_path_gpu.py -x -q -k "bench_call or odd_call or fewer" > gpurun_out/pytest_poll.txt 2>&1
tail -3 gpurun_out/pytest_poll.txt
tools/env_sweep.sh "INFV_NONE=0" "INFV_NONE=1" "INFV_POOL_ROWS=1 INFV_PR_NT=512 INFV_PR_U=2" "INFV_POOL_ROWS=1 INFV_PR_NT=256 INFV_PR_U=8" "INFV_POOL_UNROLL=8" "INFV_CHAIN_RPW=1"
cp infinite-video_amd/libinfv_ltm.so /tmp/lib_new.so; cp tools/lib_atomic.so infinite-video_amd/libinfv_ltm.so
tools/env_sweep.sh "INFV_ATOMIC=0" "INFV_ATOMIC=1" "INFV_ATOMIC=1 INFV_CHAIN_RPW=1"
cp /tmp/lib_new.so infinite-video_amd/libinfv_ltm.so
