#!/bin/bash
# single wave-priority changes again, clean experiments build: alpha kernel, UC kernel, role S without its priority
export INFV_LTM_LIBRARY=exp
for rep in 1 2; do
tools/env_sweep.sh "INFV_NONE=0" "INFV_ALPHA_PRIO=1" "INFV_UC_PRIO=1" "INFV_ALPHA_PRIO=1 INFV_UC_PRIO=1" "INFV_S_FLAGS=1"
done 2>&1 | tee gpurun_out/sweep_r05h.txt
