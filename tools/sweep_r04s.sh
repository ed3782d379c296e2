#!/bin/bash
# three against five rotating workspace sets (compile-time INFV_PSETS), same box, alternating
for rep in 1 2 3; do
for n in 3 5; do
INFV_LTM_LIBRARY=$PWD/tools/ab/lib_psets$n.so tools/env_sweep.sh "INFV_PSETS_BUILD=$n"
done; done 2>&1 | tee gpurun_out/sweep_r04s.txt
