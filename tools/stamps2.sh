#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # the INFV_* knobs below only exist in the experiments build (csrc/knobs.h)
echo "== in situ"; INFV_CHAIN_STAMPS=1 python tools/one_pass.py 2048 3 2>&1 | grep -E "batch-S stamps|pass 2" | tail -4
echo "== chain alone"; INFV_SKIP=7 INFV_CHAIN_STAMPS=1 python tools/one_pass.py 2048 3 2>&1 | grep -E "batch-S stamps|pass 2" | tail -3
tools/env_sweep.sh "INFV_NONE=0" "INFV_NONE=1"
python -m pytest tests/test_timed_path_gpu.py -x -q -k "bench_call" 2>&1 | tail -2
