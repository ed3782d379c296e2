#!/bin/bash
# projection GEMM whose matrix waves idle off the VALU port while the matrix pipe works (s_nop N after every MFMA: 4 (N + 1) clocks
# of the MFMA's 64), compile-time variants, same box alternating
for rep in 1 2; do
for n in 0 7 11 13; do
INFV_LTM_LIBRARY=$PWD/tools/ab/lib_nop$n.so tools/env_sweep.sh "INFV_GEMM_NOP_BUILD=$n"
done; done 2>&1 | tee gpurun_out/sweep_r04t.txt
for n in 11 13; do
echo "== residency, s_nop $n" | tee -a gpurun_out/sweep_r04t.txt
INFV_LTM_LIBRARY=$PWD/tools/ab/lib_nop$n.so INFV_WG_STAMPS=1 python tools/residency.py nop$n 2>&1 | grep -v amdgpu.ids | tail -18 | tee -a gpurun_out/sweep_r04t.txt
done
