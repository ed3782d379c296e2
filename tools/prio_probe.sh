#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # the INFV_* knobs below only exist in the experiments build (csrc/knobs.h)
# stream-priority experiments on the fused pooling kernel (ms of the last of three passes)
run() { echo "== $*"; env "$@" python tools/one_pass.py 2048 3 2>&1 | tail -1; }
B="INFV_PR_NT=256 INFV_PR_U=8"
run $B
run $B INFV_PRIO_POOL=-1
run $B INFV_PRIO_POOL=0
run $B INFV_PRIO_POOL=-1 ONE_PASS_PRIO=-1
run $B INFV_PRIO_POOL=-1 INFV_PRIO_UCS=-1 ONE_PASS_PRIO=-1
run $B INFV_PRIO_POOL=0 INFV_PRIO_SIDE=0
run $B INFV_PRIO_POOL=-1 INFV_PRIO_SIDE=0 ONE_PASS_PRIO=-1
run $B INFV_PRIO_POOL=0 INFV_PRIO_SIDE=1 INFV_PRIO_UCS=1 ONE_PASS_PRIO=-1
run INFV_PRIO_POOL=-1
run INFV_PR_NT=512 INFV_PR_U=8 INFV_PRIO_POOL=-1
run INFV_POOL_ROWS=0
run INFV_POOL_ROWS=0 INFV_PRIO_POOL=-1
