#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # the INFV_* knobs below only exist in the experiments build (csrc/knobs.h)
# A/B on one box: default library vs GEMM capped at 126 registers (two GEMM workgroups, or GEMM + more neighbours, per CU)
tools/env_sweep.sh "INFV_NONE=0" "INFV_NONE=1" "INFV_SUB_BATCH=36" "INFV_SUB_BATCH=48"
cp infinite-video_amd/libinfv_ltm.so /tmp/lib_w2.so; cp tools/lib_w4.so infinite-video_amd/libinfv_ltm.so
tools/env_sweep.sh "INFV_W4=1" "INFV_W4=2" "INFV_W4=1 INFV_POOL_ROWS=1 INFV_PR_NT=256 INFV_PR_U=8"
cp /tmp/lib_w2.so infinite-video_amd/libinfv_ltm.so
