#!/bin/bash
# Round-5 evidence: same-box A/B of the resident (call-long) forms against the shipped pipeline, and the residency matrix of each
# (tools/residency.py: who shares a CU with whom, time-weighted residents).  Writes gpurun_out/r05_matrix.txt, r05_residency.txt.
export PYTHONUNBUFFERED=1
one() { local label=$1; shift; echo -n "$label : "
  env "$@" timeout 300 python tools/one_pass.py ${CHUNKS:-2048} 6 2>&1 | grep "^pass" | tail -4 | awk '{print $3}' | sort -n | tr '\n' ' '; echo; }
{
echo "# ms per 2048-chunk pass (best .. worst of the last four of six), one box; then the 256-chunk shard"
for r in 1 2; do
one "shipped library                                     " INFV_LTM_LIBRARY=
one "experiments build, no knob                          " INFV_LTM_LIBRARY=exp
one "  split3_rows_kernel instead of planes by the pool  " INFV_LTM_LIBRARY=exp INFV_POOL_PLANES=0
one "call-long role S (atomics)                          " INFV_LTM_LIBRARY=exp INFV_CHAIN_CALL=1
one "call-long role S (sc1 mailboxes, linear grid)       " INFV_LTM_LIBRARY=exp INFV_CHAIN_CALL=1 INFV_CHAIN_XCD=1 INFV_CHAIN_LINEAR=1
one "call-long role S (XCD-local mailboxes)              " INFV_LTM_LIBRARY=exp INFV_CHAIN_CALL=1 INFV_CHAIN_XCD=1
one "call-long role S + pooling                          " INFV_LTM_LIBRARY=exp INFV_CHAIN_CALL=1 INFV_POOL_CALL=1
one "call-long role S + pooling + GEMM (24 workgroups)   " INFV_LTM_LIBRARY=exp INFV_CHAIN_CALL=1 INFV_POOL_CALL=1 INFV_GEMM_CALL=1 INFV_GEMM_WGS=24
one "call-long role S + pooling + GEMM (32 workgroups)   " INFV_LTM_LIBRARY=exp INFV_CHAIN_CALL=1 INFV_POOL_CALL=1 INFV_GEMM_CALL=1
one "call-long role S + pooling + GEMM (40 workgroups)   " INFV_LTM_LIBRARY=exp INFV_CHAIN_CALL=1 INFV_POOL_CALL=1 INFV_GEMM_CALL=1 INFV_GEMM_WGS=40
done
CHUNKS=256 one "256 chunks: shipped library                         " INFV_LTM_LIBRARY=
CHUNKS=256 one "256 chunks: call-long role S                        " INFV_LTM_LIBRARY=exp INFV_CHAIN_CALL=1
CHUNKS=256 one "256 chunks: call-long role S + pooling + GEMM (32)  " INFV_LTM_LIBRARY=exp INFV_CHAIN_CALL=1 INFV_POOL_CALL=1 INFV_GEMM_CALL=1
CHUNKS=256 one "256 chunks: call-long role S + pooling + GEMM (56)  " INFV_LTM_LIBRARY=exp INFV_CHAIN_CALL=1 INFV_POOL_CALL=1 INFV_GEMM_CALL=1 INFV_GEMM_WGS=56
echo "# chain only (INFV_SKIP=7: no pooling / GEMM / UC launches; garbage inputs, real timing of role S)"
one "chain only: one launch per sub-batch                " INFV_LTM_LIBRARY=exp INFV_SKIP=7
one "chain only: call-long, atomics                      " INFV_LTM_LIBRARY=exp INFV_SKIP=7 INFV_CHAIN_CALL=1
one "chain only: call-long, sc1 mailboxes                " INFV_LTM_LIBRARY=exp INFV_SKIP=7 INFV_CHAIN_CALL=1 INFV_CHAIN_XCD=1 INFV_CHAIN_LINEAR=1
one "chain only: call-long, XCD-local mailboxes          " INFV_LTM_LIBRARY=exp INFV_SKIP=7 INFV_CHAIN_CALL=1 INFV_CHAIN_XCD=1
echo "# which stage hides how much (shipped pipeline; INFV_SKIP 16 = no alpha launches, 32 = no UC, 4 = neither, 2 = no GEMM)"
for m in 0 16 32 4 2; do one "INFV_SKIP=$m                                         " INFV_LTM_LIBRARY=exp INFV_SKIP=$m; done
} > gpurun_out/r05_matrix.txt 2>&1
{
res() { echo "== $1"; shift; env INFV_LTM_LIBRARY=exp INFV_WG_STAMPS=1 "$@" timeout 300 python tools/residency.py r05 2>&1 | grep -v amdgpu.ids | tail -13; }
res "shipped pipeline (one launch per sub-batch)"
res "call-long role S" INFV_CHAIN_CALL=1
res "call-long role S + pooling" INFV_CHAIN_CALL=1 INFV_POOL_CALL=1
res "call-long role S + pooling + GEMM (32 workgroups; the resident GEMM is not stamped: its 32 CUs count as 'no workgroup of these kernels')" INFV_CHAIN_CALL=1 INFV_POOL_CALL=1 INFV_GEMM_CALL=1
res "shipped pipeline without the UC and alpha launches (INFV_SKIP=4)" INFV_SKIP=4
} > gpurun_out/r05_residency.txt 2>&1
tail -5 gpurun_out/r05_matrix.txt; tail -4 gpurun_out/r05_residency.txt
