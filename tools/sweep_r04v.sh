#!/bin/bash
# (the "logical" library tools/ab/lib_uc_old.so was built from the previous commit's ltm_uc.hip; the parity-order variant was not kept)
# UC kernel with even boxes first in LDS (parity row order) against the logical row order, same box alternating
{
python -m pytest tests/test_ltm_gpu.py tests/test_timed_path_gpu.py -x -q 2>&1 | tail -2
for rep in 1 2 3; do
INFV_LTM_LIBRARY=$PWD/tools/ab/lib_uc_old.so tools/env_sweep.sh "UC_ROWS=logical"
INFV_LTM_LIBRARY=exp tools/env_sweep.sh "UC_ROWS=parity"
done
} 2>&1 | tee gpurun_out/sweep_r04v.txt
