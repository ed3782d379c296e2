#!/bin/bash
# (the libraries tools/ab/lib_aux<N>.so were built from ltm_kernels.hip with the aux constant of TokF32::load_nt changed; not kept)
# cache-policy bits of the pooling kernel's buffer loads (aux: 1 sc0, 2 nt, 16 sc1; the shipped kernel uses 2): libraries built with the
# constant changed (tools/ab/lib_aux<N>.so), same box, and the GEMM's / UC kernel's HBM fetch beside each
{
for rep in 1 2; do
for a in 2 0 1 3 16 17 18 19; do
if [ $a = 2 ]; then lib=exp; else lib=$PWD/tools/ab/lib_aux$a.so; fi
INFV_LTM_LIBRARY=$lib tools/env_sweep.sh "POOL_LOAD_AUX=$a"
done; done
} 2>&1 | tee gpurun_out/sweep_r05g.txt
