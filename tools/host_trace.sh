export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # the INFV_* knobs below only exist in the experiments build (csrc/knobs.h)
export INFV_HOST_TRACE=1
echo "== new 256/8"; INFV_PR_NT=256 INFV_PR_U=8 python tools/one_pass.py 2048 3 2>&1 | tail -4
echo "== old"; INFV_POOL_ROWS=0 python tools/one_pass.py 2048 3 2>&1 | tail -4
