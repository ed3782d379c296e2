#!/bin/bash
# Every one-off A/B sweep of rounds 3-4 as it was run on the GPU box (results: profiles/r03_sweeps/sweep_<name>.txt), in ONE
# parameterised script:   tools/sweeps.sh <name>     (names: r03a ... r05e; `tools/sweeps.sh list` prints them).
# A sweep is a list of environments for tools/env_sweep.sh (the short headline bench under each) plus, for some, residency
# stamps or traces.  They are the record of how each result file was produced: several use knobs or kernels that were removed
# after the measurement (INFV_WHOLE_CALL, INFV_GEMM_X, INFV_POOL_ROWS=1, INFV_POOL_DB, INFV_PROJ_FP32, ...) and no longer run as written.
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # the INFV_* knobs only exist in the experiments build (csrc/knobs.h)
name=$1
case "$name" in
r03a)
# round-3 first sweep: which kernels can share a CU (registers / LDS / wave slots) decides the pooling rate in situ
tools/env_sweep.sh \
 "INFV_NONE=0" \
 "INFV_WHOLE_CALL=1" \
 "INFV_POOL_UNROLL=16" \
 "INFV_POOL_UNROLL=8" \
 "INFV_WHOLE_CALL=1 INFV_POOL_UNROLL=16" \
 "INFV_POOL_PAD=1 INFV_POOL_UNROLL=2" \
 "INFV_POOL_PAD=1 INFV_POOL_UNROLL=4" \
 "INFV_WHOLE_CALL=1 INFV_POOL_PAD=1 INFV_POOL_UNROLL=2" \
 "INFV_WHOLE_CALL=1 INFV_POOL_PAD=1 INFV_POOL_UNROLL=4" \
 "INFV_WHOLE_CALL=1 INFV_POOL_PAD=1 INFV_POOL_UNROLL=8" \
 "INFV_GEMM_LW=0" \
 "INFV_WHOLE_CALL=1 INFV_GEMM_LW=0 INFV_POOL_PAD=1 INFV_POOL_UNROLL=4" \
 "INFV_WHOLE_CALL=1 INFV_GEMM_LW=0 INFV_POOL_PAD=40000 INFV_POOL_UNROLL=8" \
 "INFV_VPROJ_ON_UC=0" \
 "INFV_WHOLE_CALL=1 INFV_VPROJ_ON_UC=0 INFV_POOL_PAD=1 INFV_POOL_UNROLL=4" \
 2>&1 | tee gpurun_out/sweep_r03a.txt
;;
r03b)
# round-3 sweep b: the fused pool+rows kernel, its footprint (threads / loads per group / padding LDS) and who may sit beside it
tools/env_sweep.sh \
 "INFV_POOL_ROWS=0" \
 "INFV_NONE=0" \
 "INFV_PR_NT=256 INFV_PR_U=4 INFV_PR_PAD=0" \
 "INFV_PR_NT=256 INFV_PR_U=4 INFV_PR_PAD=0 INFV_S_LDS=150000" \
 "INFV_PR_NT=256 INFV_PR_U=8 INFV_PR_PAD=0" \
 "INFV_PR_NT=256 INFV_PR_U=4 INFV_PR_PAD=13000 INFV_S_LDS=150000" \
 "INFV_PR_NT=256 INFV_PR_U=4 INFV_PR_PAD=0 INFV_S_LDS=150000 INFV_WHOLE_CALL=1" \
 "INFV_PR_NT=256 INFV_PR_U=4 INFV_PR_PAD=40000" \
 "INFV_PR_NT=256 INFV_PR_U=4 INFV_PR_PAD=40000 INFV_S_LDS=150000" \
 "INFV_PR_NT=256 INFV_PR_U=8 INFV_PR_PAD=40000" \
 "INFV_PR_NT=512 INFV_PR_U=4 INFV_PR_PAD=60000" \
 "INFV_PR_NT=512 INFV_PR_U=8" \
 "INFV_PR_NT=256 INFV_PR_U=2 INFV_PR_PAD=0 INFV_S_LDS=150000" \
 "INFV_PR_NT=256 INFV_PR_U=8 INFV_PR_PAD=84000" \
 "INFV_PR_NT=256 INFV_PR_U=8 INFV_PR_PAD=60000" \
 2>&1 | tee gpurun_out/sweep_r03b.txt
;;
r03c)
# round-3 sweep c: 8 R sets; bounded (grid-stride) pooling grids with a small LDS pad that keeps them off the chain's CUs
tools/env_sweep.sh \
 "INFV_POOL_ROWS=0" \
 "INFV_NONE=0" \
 "INFV_PR_NT=256 INFV_PR_U=8" \
 "INFV_PR_NT=512 INFV_PR_U=8" \
 "INFV_PR_NT=256 INFV_PR_U=8 INFV_PR_WGS=256" \
 "INFV_PR_NT=256 INFV_PR_U=4 INFV_PR_PAD=12000 INFV_S_LDS=150000 INFV_PR_WGS=320" \
 "INFV_PR_NT=256 INFV_PR_U=4 INFV_PR_PAD=12000 INFV_S_LDS=150000 INFV_PR_WGS=480" \
 "INFV_PR_NT=256 INFV_PR_U=4 INFV_PR_PAD=12000 INFV_S_LDS=150000 INFV_PR_WGS=640" \
 "INFV_PR_NT=256 INFV_PR_U=8 INFV_PR_PAD=12000 INFV_S_LDS=150000 INFV_PR_WGS=320" \
 "INFV_PR_NT=256 INFV_PR_U=8 INFV_PR_PAD=12000 INFV_S_LDS=150000 INFV_PR_WGS=480" \
 "INFV_PR_NT=512 INFV_PR_U=4 INFV_PR_PAD=12000 INFV_S_LDS=150000 INFV_PR_WGS=256" \
 "INFV_PR_NT=512 INFV_PR_U=8 INFV_PR_PAD=12000 INFV_S_LDS=150000 INFV_PR_WGS=256" \
 "INFV_PR_NT=256 INFV_PR_U=4 INFV_PR_PAD=0 INFV_PR_WGS=480" \
 "INFV_PR_NT=256 INFV_PR_U=8 INFV_PR_PAD=0 INFV_PR_WGS=320" \
 2>&1 | tee gpurun_out/sweep_r03c.txt
;;
r03d)
tools/env_sweep.sh \
 "INFV_PR_NT=256 INFV_PR_U=8 INFV_VPROJ_ON_UC=0" \
 "INFV_PR_NT=512 INFV_PR_U=8 INFV_VPROJ_ON_UC=0" \
 "INFV_PR_NT=512 INFV_PR_U=4 INFV_VPROJ_ON_UC=0" \
 "INFV_PR_NT=256 INFV_PR_U=4 INFV_VPROJ_ON_UC=0" \
 "INFV_PR_NT=256 INFV_PR_U=8 INFV_VPROJ_ON_UC=0 INFV_SUB_BATCH=32" \
 "INFV_PR_NT=256 INFV_PR_U=8 INFV_VPROJ_ON_UC=0 INFV_PR_PAD=60000" \
 "INFV_PR_NT=256 INFV_PR_U=8 INFV_VPROJ_ON_UC=0 INFV_PR_PAD=100000" \
 "INFV_PR_NT=256 INFV_PR_U=8 INFV_VPROJ_ON_UC=0 INFV_GEMM_LW=0" \
 "INFV_POOL_ROWS=0 INFV_VPROJ_ON_UC=0" \
 2>&1 | tee gpurun_out/sweep_r03d.txt
;;
r03e)
tools/env_sweep.sh \
 "INFV_POOL_ROWS=0 INFV_CHAIN_RPW=1" \
 "INFV_POOL_ROWS=0" \
 "INFV_POOL_ROWS=0 INFV_VPROJ_ON_UC=0" \
 "INFV_PR_NT=256 INFV_PR_U=8" \
 "INFV_PR_NT=256 INFV_PR_U=8 INFV_VPROJ_ON_UC=0" \
 "INFV_PR_NT=512 INFV_PR_U=4" \
 "INFV_PR_NT=512 INFV_PR_U=8" \
 "INFV_POOL_ROWS=0 INFV_WHOLE_CALL=1" \
 "INFV_PR_NT=256 INFV_PR_U=8 INFV_WHOLE_CALL=1" \
 2>&1 | tee gpurun_out/sweep_r03e.txt
;;
r03f)
export INFV_VPROJ_ON_UC=0
tools/env_sweep.sh \
 "INFV_POOL_ROWS=0" \
 "INFV_POOL_ROWS=0 INFV_POOL_UNROLL=8" \
 "INFV_POOL_ROWS=0 INFV_POOL_NT=1024" \
 "INFV_POOL_ROWS=0 INFV_POOL_NT=1024 INFV_POOL_UNROLL=2" \
 "INFV_PR_NT=512 INFV_PR_U=2" \
 "INFV_PR_NT=512 INFV_PR_U=4" \
 "INFV_PR_NT=256 INFV_PR_U=4" \
 "INFV_PR_NT=256 INFV_PR_U=2" \
 "INFV_PR_NT=256 INFV_PR_U=4 INFV_PR_PAD=50000" \
 "INFV_PR_NT=256 INFV_PR_U=2 INFV_PR_PAD=50000" \
 "INFV_POOL_ROWS=0 INFV_SUB_BATCH=32" \
 "INFV_POOL_ROWS=0 INFV_CHAIN_RPW=1" \
 2>&1 | tee gpurun_out/sweep_r03f.txt
;;
r03g)
export INFV_VPROJ_ON_UC=0 INFV_POOL_ROWS=0
tools/env_sweep.sh \
 "INFV_NONE=0" \
 "INFV_WHOLE_CALL=1" \
 "INFV_WHOLE_CALL=1 INFV_POOL_UNROLL=8" \
 "INFV_WHOLE_CALL=1 INFV_SUB_BATCH=32" \
 "INFV_WHOLE_CALL=1 INFV_SUB_BATCH=21" \
 "INFV_WHOLE_CALL=1 INFV_POOL_ROWS=1 INFV_PR_NT=512 INFV_PR_U=2" \
 2>&1 | tee gpurun_out/sweep_r03g.txt
;;
r03h)
export INFV_VPROJ_ON_UC=0 INFV_POOL_ROWS=0
tools/env_sweep.sh \
 "INFV_POOL_PAD=86016" \
 "INFV_POOL_PAD=76000" \
 "INFV_POOL_PAD=64000" \
 "INFV_POOL_PAD=56000" \
 "INFV_POOL_PAD=48000" \
 "INFV_POOL_PAD=56000 INFV_POOL_UNROLL=2" \
 "INFV_POOL_PAD=40000 INFV_POOL_UNROLL=2" \
 "INFV_POOL_PAD=56000 INFV_POOL_UNROLL=8" \
 "INFV_POOL_PAD=100000" \
 2>&1 | tee gpurun_out/sweep_r03h.txt
;;
r03i)
export INFV_VPROJ_ON_UC=0 INFV_POOL_ROWS=0
tools/env_sweep.sh \
 "INFV_NONE=0" \
 "INFV_WHOLE_CALL=1 INFV_POOL_PAD=56000 INFV_POOL_UNROLL=2" \
 "INFV_WHOLE_CALL=1 INFV_POOL_PAD=56000 INFV_POOL_UNROLL=4" \
 "INFV_WHOLE_CALL=1 INFV_POOL_PAD=40000 INFV_POOL_UNROLL=2" \
 "INFV_WHOLE_CALL=1 INFV_POOL_PAD=40000 INFV_POOL_UNROLL=4" \
 "INFV_WHOLE_CALL=1 INFV_POOL_PAD=1 INFV_POOL_UNROLL=2" \
 "INFV_WHOLE_CALL=1 INFV_POOL_PAD=64000 INFV_POOL_UNROLL=4" \
 "INFV_WHOLE_CALL=1 INFV_POOL_ROWS=1 INFV_PR_NT=256 INFV_PR_U=4 INFV_PR_PAD=56000" \
 "INFV_WHOLE_CALL=1 INFV_POOL_ROWS=1 INFV_PR_NT=512 INFV_PR_U=2 INFV_PR_PAD=56000" \
 "INFV_WHOLE_CALL=1 INFV_POOL_ROWS=1 INFV_PR_NT=256 INFV_PR_U=8 INFV_PR_PAD=40000" \
 2>&1 | tee gpurun_out/sweep_r03i.txt
;;
r03k)
tools/env_sweep.sh \
 "INFV_NONE=0" \
 "INFV_POOL_NT=1024 INFV_POOL_UNROLL=2 INFV_POOL_PAD=1" \
 "INFV_POOL_NT=1024 INFV_POOL_UNROLL=4 INFV_POOL_PAD=1" \
 "INFV_POOL_NT=1024 INFV_POOL_UNROLL=2 INFV_POOL_PAD=30000" \
 "INFV_POOL_NT=1024 INFV_POOL_UNROLL=4 INFV_POOL_PAD=30000" \
 "INFV_POOL_NT=1024 INFV_POOL_UNROLL=2 INFV_POOL_PAD=60000" \
 "INFV_POOL_NT=1024 INFV_POOL_UNROLL=4 INFV_POOL_PAD=60000" \
 "INFV_POOL_NT=1024 INFV_POOL_UNROLL=4" \
 "INFV_NONE=1" \
 2>&1 | tee gpurun_out/sweep_r03k.txt
;;
r03l)
tools/env_sweep.sh "INFV_NONE=0" "INFV_POOL_DB=1" "INFV_POOL_DB=2" "INFV_POOL_DB=4" "INFV_NONE=1" "INFV_POOL_DB=2 INFV_POOL_PAD=100000" "INFV_POOL_DB=1 INFV_POOL_NT=512" 2>&1 | tee gpurun_out/sweep_r03l.txt
python -m pytest tests/test_ltm_gpu.py -x -q -k "pool or pieces or pooled" 2>&1 | tail -2
;;
r03m)
# who slows the pooling stream (16-row chain tiles, V' projection on the side stream): INFV_SKIP 2 = no GEMM, 4 = no UC (+alpha), 8 = no chain
tools/env_sweep.sh "INFV_SKIP=0" "INFV_SKIP=2" "INFV_SKIP=4" "INFV_SKIP=6" "INFV_SKIP=8" "INFV_SKIP=10" "INFV_SKIP=12" "INFV_SKIP=14" \
  "INFV_SKIP=6 INFV_POOL_UNROLL=16" "INFV_SKIP=14 INFV_POOL_UNROLL=16" 2>&1 | tee gpurun_out/sweep_r03m.txt
;;
r03n)
# pooling workgroups small enough to share a CU with role S (four waves, <= 72 registers)
tools/env_sweep.sh "INFV_NONE=0" \
 "INFV_POOL_NT=256 INFV_POOL_UNROLL=8" \
 "INFV_POOL_NT=256 INFV_POOL_UNROLL=4" \
 "INFV_POOL_NT=256 INFV_POOL_UNROLL=8 INFV_POOL_PAD=60000" \
 "INFV_POOL_NT=256 INFV_POOL_UNROLL=8 INFV_POOL_PAD=44000" \
 "INFV_POOL_NT=256 INFV_POOL_UNROLL=4 INFV_POOL_PAD=44000" \
 "INFV_POOL_NT=256 INFV_POOL_UNROLL=8 INFV_POOL_PAD=30000" \
 "INFV_POOL_NT=256 INFV_POOL_UNROLL=4 INFV_POOL_PAD=30000" \
 "INFV_NONE=1" 2>&1 | tee gpurun_out/sweep_r03n.txt
;;
r03o)
# sub-batch size with the round-3 structure (fewer role-S launches per video vs longer pipeline fill)
for bc in 42 56 63 84 42 28; do
  echo -n "batch-chunks $bc: "
  python bench.py --steps 8 --warmup 2 --batch-chunks $bc --no-cpu-baseline --no-encode-video --no-selfcheck --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_pass']
print(round(d['value']), 'chunks/s wall', round(d['ms_per_step'],2), 'pool', k['pool'], 'project', k['project'], 'chain', k['chain'], 'uc', k['uc'])"
done 2>&1 | tee gpurun_out/sweep_r03o.txt
;;
r03p)
# pooling + projection GEMM only (no role S, no UC): how do the two share CUs?
export INFV_SKIP=12
tools/env_sweep.sh "INFV_NONE=0" \
 "INFV_GEMM_LW=0" \
 "INFV_POOL_UNROLL=16" \
 "INFV_POOL_UNROLL=2" \
 "INFV_POOL_NT=1024" \
 "INFV_POOL_PAD=1 INFV_POOL_UNROLL=4" \
 "INFV_POOL_PAD=1 INFV_POOL_UNROLL=16" \
 "INFV_GEMM_PAD=90000" \
 "INFV_GEMM_LW=0 INFV_GEMM_PAD=90000" \
 "INFV_GEMM_LW=0 INFV_POOL_PAD=1 INFV_POOL_UNROLL=8" \
 2>&1 | tee gpurun_out/sweep_r03p.txt
;;
r03q)
tools/env_sweep.sh "INFV_NONE=0" "INFV_GEMM_LW=0" "INFV_NONE=1" "INFV_GEMM_LW=0 INFV_POOL_UNROLL=8" "INFV_GEMM_LW=0 INFV_POOL_PAD=60000" 2>&1 | tee gpurun_out/sweep_r03q.txt
;;
r03r)
# pool_rows2_kernel (fused pool + rows, short-lived workgroups) against the default two-kernel pooling, same box
{
INFV_POOL_ROWS=2 tools/quick_bench.sh pr2 6
tools/env_sweep.sh "INFV_NONE=0" "INFV_POOL_ROWS=2" "INFV_POOL_ROWS=2 INFV_PR_U=8" "INFV_POOL_ROWS=2 INFV_PR_PAD=60000" "INFV_POOL_ROWS=2 INFV_PR_U=2" "INFV_NONE=1"
} 2>&1 | tee gpurun_out/sweep_r03r.txt
;;
r03s)
# alternating A/B on one box: default pooling (pool_frames + build_rows) vs pool_rows2_kernel with 8-load bursts
tools/env_sweep.sh "INFV_NONE=0" "INFV_POOL_ROWS=2 INFV_PR_U=8" "INFV_NONE=1" "INFV_POOL_ROWS=2 INFV_PR_U=8" "INFV_NONE=2" "INFV_POOL_ROWS=2 INFV_PR_U=8" "INFV_NONE=3" "INFV_POOL_ROWS=2 INFV_PR_U=8" "INFV_POOL_ROWS=2 INFV_PR_U=4" "INFV_POOL_ROWS=2 INFV_PR_U=4" 2>&1 | tee gpurun_out/sweep_r03s.txt
;;
r03t)
# per-CU streaming rate of the pooling kernel alone, against the number of CUs it may use
{
for w in 32 64 128 192 208 256; do INFV_PR_WGS=$w python tools/pool_cus.py 2>/dev/null | tail -1; done
for w in 256 416 512; do INFV_PR_WGS=$w INFV_PR_PAD=40000 python tools/pool_cus.py 2>/dev/null | tail -1; done
python tools/pool_cus.py 2>/dev/null | tail -1
} | tee gpurun_out/sweep_r03t.txt
;;
r03u)
# residency of the pooling workgroups: in situ, alone, and beside each of the other three streams
{
echo "== in situ"; RES_TAG=insitu python tools/residency.py 2>&1 | tail -14
echo "== pooling alone (INFV_SKIP=14: everything else skipped)"; RES_TAG=alone INFV_SKIP=14 python tools/residency.py 2>&1 | tail -14
echo "== pooling + role S (INFV_SKIP=6)"; RES_TAG=chain INFV_SKIP=6 python tools/residency.py 2>&1 | tail -14
echo "== pooling + GEMM (INFV_SKIP=12)"; RES_TAG=gemm INFV_SKIP=12 python tools/residency.py 2>&1 | tail -14
echo "== pooling + UC (INFV_SKIP=10)"; RES_TAG=uc INFV_SKIP=10 python tools/residency.py 2>&1 | tail -14
} | tee gpurun_out/sweep_r03u.txt
;;
r03v)
# does the pooling workgroup share a CU with a UC workgroup?  register footprint 3 x 64 (U=8) against 3 x 56 (U=2) per SIMD
{
echo "== pooling + UC, U=8"; INFV_SKIP=10 python tools/residency.py 2>&1 | tail -13 | head -4
echo "== pooling + UC, U=2"; INFV_PR_U=2 INFV_SKIP=10 python tools/residency.py 2>&1 | tail -13 | head -4
echo "== pooling + UC, U=2, pad 80K"; INFV_PR_PAD=81920 INFV_PR_U=2 INFV_SKIP=10 python tools/residency.py 2>&1 | tail -13 | head -4
echo "== pooling + UC, U=8, pad 80K"; INFV_PR_PAD=81920 INFV_SKIP=10 python tools/residency.py 2>&1 | tail -13 | head -4
echo "== in situ U=2"; INFV_PR_U=2 python tools/residency.py 2>&1 | tail -13 | head -4
echo "== in situ U=2, pad 80K"; INFV_PR_PAD=81920 INFV_PR_U=2 python tools/residency.py 2>&1 | tail -13 | head -4
} | tee gpurun_out/sweep_r03v.txt
;;
r03w)
# pool_rows2 on scalar-resource buffer loads (56 VGPRs at U=8, 40 at U=4: shares a SIMD with UC waves): wall clock per setting
tools/env_sweep.sh "INFV_NONE=0" "INFV_PR_U=4" "INFV_PR_PAD=81920" "INFV_PR_U=4 INFV_PR_PAD=81920" "INFV_NONE=1" "INFV_PR_U=4" "INFV_PR_PAD=81920" "INFV_PR_U=4 INFV_PR_PAD=81920" "INFV_PR_PAD=65536" "INFV_POOL_ROWS=0" 2>&1 | tee gpurun_out/sweep_r03w.txt
echo "== residency in situ (U=8, 84K)" | tee -a gpurun_out/sweep_r03w.txt
INFV_WG_STAMPS=1 python tools/residency.py 2>&1 | tail -13 | head -4 | tee -a gpurun_out/sweep_r03w.txt
;;
r03x)
# residency of the pooling workgroups with the 40-VGPR instantiation (U=4), in situ and beside UC / GEMM only
{
echo "== in situ U=4"; python tools/residency.py 2>&1 | tail -13
echo "== pooling + UC U=4"; INFV_SKIP=10 python tools/residency.py 2>&1 | tail -13 | head -4
echo "== pooling + GEMM U=4"; INFV_SKIP=12 python tools/residency.py 2>&1 | tail -13 | head -4
echo "== pooling + UC + GEMM U=4 (no chain)"; INFV_SKIP=8 python tools/residency.py 2>&1 | tail -13 | head -4
echo "== alone U=4"; INFV_SKIP=14 python tools/residency.py 2>&1 | tail -13 | head -4
echo "== alone U=8"; INFV_PR_U=8 INFV_SKIP=14 python tools/residency.py 2>&1 | tail -13 | head -4
} | tee gpurun_out/sweep_r03x.txt
;;
r03y)
# what in the projection GEMM stretches a co-resident pooling workgroup?  pooling + GEMM only, GEMM variants (timing only):
# X=1 one MFMA per tile, X=2 loaders move nothing, X=4 MFMA waves sleep instead, X=6 sleep + no loads (LDS reads and barriers only)
{
for x in 0 1 2 3 4 6; do echo "== pooling + GEMM, INFV_GEMM_X=$x"; INFV_GEMM_X=$x python tools/residency.py 2>&1 | tail -13 | head -4; done
} | tee gpurun_out/sweep_r03y.txt
;;
r03z)
# s_setprio(3) in the pooling kernel: lifetime of its workgroups beside the GEMM, then wall clock in situ
{
for pr in 0 1; do echo "== pooling + GEMM, U=4, INFV_POOL_PRIO=$pr"; INFV_WG_STAMPS=1 INFV_PR_U=4 INFV_SKIP=12 INFV_POOL_PRIO=$pr python tools/residency.py 2>&1 | tail -13 | head -4; done
for pr in 0 1; do echo "== in situ, U=4, INFV_POOL_PRIO=$pr"; INFV_WG_STAMPS=1 INFV_PR_U=4 INFV_POOL_PRIO=$pr python tools/residency.py 2>&1 | tail -13 | head -4; done
tools/env_sweep.sh "INFV_POOL_PRIO=0 INFV_PR_U=4" "INFV_POOL_PRIO=1 INFV_PR_U=4" "INFV_POOL_PRIO=0 INFV_PR_U=8" "INFV_POOL_PRIO=1 INFV_PR_U=8" "INFV_POOL_PRIO=0 INFV_PR_U=4" "INFV_POOL_PRIO=1 INFV_PR_U=4" "INFV_POOL_PRIO=1 INFV_PR_U=4 INFV_PR_PAD=81920" "INFV_POOL_PRIO=1 INFV_PR_U=8 INFV_PR_PAD=81920"
} 2>&1 | tee gpurun_out/sweep_r03z.txt
;;
r04a)
# who shares a CU with whom: residency of every pipeline kernel's workgroups in one headline pass
{
python tools/residency.py insitu_u8 2>&1 | grep -v amdgpu.ids | tail -18
INFV_PR_U=4 python tools/residency.py insitu_u4 2>&1 | grep -v amdgpu.ids | tail -18
} | tee gpurun_out/sweep_r04a.txt
;;
r04b)
# wave priorities: pooling above the GEMM's matrix waves, UC above the pooling
{
tools/env_sweep.sh "INFV_NONE=0" "INFV_POOL_PRIO=1" "INFV_POOL_PRIO=1 INFV_UC_PRIO=2" "INFV_POOL_PRIO=2 INFV_UC_PRIO=3" "INFV_NONE=1" "INFV_POOL_PRIO=1" "INFV_POOL_PRIO=1 INFV_UC_PRIO=2" "INFV_POOL_PRIO=3 INFV_UC_PRIO=3"
INFV_WG_STAMPS=1 INFV_POOL_PRIO=1 INFV_UC_PRIO=2 python tools/residency.py prio12 2>&1 | grep -v amdgpu.ids | tail -18
} 2>&1 | tee gpurun_out/sweep_r04b.txt
;;
r04c)
# the GEMM's matrix waves idle in s_nop between MFMAs (X=8: 32 cycles, X=16: 48) so that co-resident waves get the VALU port
{
for x in 0 8 16; do echo "== pooling + GEMM only, INFV_GEMM_X=$x"; INFV_WG_STAMPS=1 INFV_SKIP=12 INFV_GEMM_X=$x python tools/residency.py x 2>&1 | grep -E "pool |gemm " ; done
tools/env_sweep.sh "INFV_NONE=0" "INFV_GEMM_X=8" "INFV_GEMM_X=16" "INFV_NONE=1" "INFV_GEMM_X=8" "INFV_GEMM_X=16"
INFV_WG_STAMPS=1 INFV_GEMM_X=16 python tools/residency.py x16 2>&1 | grep -v amdgpu.ids | tail -18
} 2>&1 | tee gpurun_out/sweep_r04c.txt
;;
r04d)
# wave priorities: everything above the GEMM's matrix waves (role S stays at 3)
{
tools/env_sweep.sh "INFV_NONE=0" "INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2" "INFV_POOL_PRIO=1 INFV_UC_PRIO=1 INFV_ALPHA_PRIO=1" "INFV_ALPHA_PRIO=1" "INFV_NONE=1" "INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2" "INFV_POOL_PRIO=1 INFV_UC_PRIO=1 INFV_ALPHA_PRIO=1" "INFV_ALPHA_PRIO=1"
INFV_WG_STAMPS=1 INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2 python tools/residency.py prio122 2>&1 | grep -v amdgpu.ids | tail -18
} 2>&1 | tee gpurun_out/sweep_r04d.txt
;;
r04e)
# HIP runtime hardware-queue count: do the pipeline's four streams share AQL queues / CP pipes?
{
tools/env_sweep.sh "INFV_NONE=0" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=2" "GPU_MAX_HW_QUEUES=16" "INFV_NONE=1" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=6" "GPU_MAX_HW_QUEUES=8 INFV_PR_U=4"
} 2>&1 | tee gpurun_out/sweep_r04e.txt
;;
r04f)
# pooling kernel as a fixed number of long-lived grid-stride workgroups (no re-dispatch while other kernels' backlogs hold the dispatcher)
{
tools/env_sweep.sh "INFV_NONE=0" "INFV_PR_WGS=128" "INFV_PR_WGS=160" "INFV_PR_WGS=192" "INFV_PR_WGS=208" "INFV_PR_WGS=256" "INFV_PR_WGS=160 INFV_PR_U=4" "INFV_PR_WGS=192 INFV_PR_U=4" "INFV_PR_WGS=224 INFV_PR_U=4" "INFV_NONE=1"
} 2>&1 | tee gpurun_out/sweep_r04f.txt
;;
r04g)
# does a smaller padding LDS let a pooling workgroup join a CU where a GEMM workgroup arrived first?
{
for pad in 86016 83968 82944 82000; do
echo "== pad $pad"; INFV_PR_PAD=$pad INFV_WG_STAMPS=1 python tools/residency.py pad$pad 2>&1 | grep -E "span|pool |gemm  |    gemm"
done
tools/env_sweep.sh "INFV_PR_PAD=86016" "INFV_PR_PAD=83968" "INFV_PR_PAD=82944" "INFV_PR_PAD=82000" "INFV_PR_PAD=86016" "INFV_PR_PAD=82944"
} 2>&1 | tee gpurun_out/sweep_r04g.txt
;;
r04h)
# projection GEMM launched as column slices (no dispatcher backlog), with and without the wave priorities
{
tools/env_sweep.sh "INFV_NONE=0" "INFV_GEMM_SLICES=2" "INFV_GEMM_SLICES=3" "INFV_GEMM_SLICES=2 INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2" "INFV_GEMM_SLICES=3 INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2" "INFV_NONE=1" "INFV_GEMM_SLICES=2" "INFV_GEMM_SLICES=2 INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2" "INFV_GEMM_SLICES=3 INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2"
INFV_GEMM_SLICES=2 INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2 INFV_WG_STAMPS=1 python tools/residency.py sl2prio 2>&1 | grep -v amdgpu.ids | tail -18
} 2>&1 | tee gpurun_out/sweep_r04h.txt
;;
r04i)
# which other stream stops pooling workgroups from joining CUs that hold a GEMM workgroup?  (priorities on, U=4)
{
for sk in 12 8 4; do
echo "== INFV_SKIP=$sk (1 pool, 2 GEMM, 4 UC+alpha, 8 chain)"; INFV_SKIP=$sk python tools/residency.py skip$sk 2>&1 | grep -v amdgpu.ids | tail -16
done
} 2>&1 | tee gpurun_out/sweep_r04i.txt
;;
r04j)
# longer sub-batches (fewer role-S launches) with and without long-lived pooling workgroups
run() {  # $1 = batch chunks, rest = env
  bc=$1; shift
  out=$(env "$@" python bench.py --steps 6 --warmup 2 --batch-chunks $bc --no-cpu-baseline --no-encode-video --no-selfcheck --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_pass']
print(round(d['value']), 'chunks/s wall', round(d['ms_per_step'],2), 'pool', k['pool'], 'project', k['project'], 'chain', k['chain'], 'uc', k['uc'])")
  echo "sweep [batch $bc $*] $out"
}
{
run 42 INFV_NONE=0
run 84 INFV_NONE=0
run 126 INFV_NONE=0
run 84 INFV_PR_WGS=160
run 84 INFV_PR_WGS=192
run 126 INFV_PR_WGS=160
run 126 INFV_PR_WGS=192
run 168 INFV_PR_WGS=176
run 42 INFV_NONE=1
} 2>&1 | tee gpurun_out/sweep_r04j.txt
;;
r04k)
# same-box alternating A/B of the shippable settings: burst length of the pooling kernel x column slices of the GEMM
for rep in 1 2 3; do
tools/env_sweep.sh "INFV_PR_U=8" "INFV_PR_U=4" "INFV_PR_U=8 INFV_GEMM_SLICES=2" "INFV_PR_U=4 INFV_GEMM_SLICES=2" "INFV_POOL_ROWS=0"
done 2>&1 | tee gpurun_out/sweep_r04k.txt
python - <<'PY'
import re,collections
d=collections.defaultdict(list)
for l in open("gpurun_out/sweep_r04k.txt"):
    m=re.match(r"sweep \[(.*)\] (\d+) chunks",l)
    if m: d[m.group(1)].append(int(m.group(2)))
for k,v in d.items(): print(k, v, "mean", sum(v)//len(v))
PY
;;
r04l)
# (after removing the GEMM's experiment branches, which had cost the experiments build's GEMM 202 VGPRs against the shipped 130
#  and with them its seat beside a pooling workgroup: sweeps r03y..r04k ran with that handicap)
# burst length, wave priorities and GEMM column slices again, same box, alternating
for rep in 1 2; do
tools/env_sweep.sh "INFV_PR_U=8" "INFV_PR_U=4" "INFV_PR_U=8 INFV_POOL_PRIO=1" "INFV_PR_U=4 INFV_POOL_PRIO=1" "INFV_PR_U=4 INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2" "INFV_PR_U=8 INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2" "INFV_PR_U=8 INFV_GEMM_SLICES=2" "INFV_POOL_ROWS=0"
done 2>&1 | tee gpurun_out/sweep_r04l.txt
INFV_LTM_LIBRARY= python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-encode-video --no-selfcheck --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('shipped library', round(d['value']), 'chunks/s', d['roofline']['kernel_ms_per_pass'])" | tee -a gpurun_out/sweep_r04l.txt
python - <<'PY' | tee -a gpurun_out/sweep_r04l.txt
import re,collections
d=collections.defaultdict(list)
for l in open("gpurun_out/sweep_r04l.txt"):
    m=re.match(r"sweep \[(.*)\] (\d+) chunks",l)
    if m: d[m.group(1)].append(int(m.group(2)))
for k,v in d.items(): print(k, v, "mean", sum(v)//len(v))
PY
for cfg in "INFV_PR_U=8" "INFV_PR_U=8 INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2"; do
echo "== residency [$cfg]" | tee -a gpurun_out/sweep_r04l.txt
env $cfg INFV_WG_STAMPS=1 python tools/residency.py l 2>&1 | grep -v amdgpu.ids | tail -18 | tee -a gpurun_out/sweep_r04l.txt
done
;;
r04n)
# projection GEMM with 90 KB of LDS per workgroup: no pooling workgroup beside it (the pooling keeps full speed, the GEMM its CU)
for rep in 1 2; do
tools/env_sweep.sh "INFV_NONE=0" "INFV_GEMM_LW_LDS=92160" "INFV_GEMM_LW_LDS=92160 INFV_PR_U=4" "INFV_GEMM_LW_LDS=98304"
done 2>&1 | tee gpurun_out/sweep_r04n.txt
INFV_GEMM_LW_LDS=92160 INFV_WG_STAMPS=1 python tools/residency.py lds90 2>&1 | grep -v amdgpu.ids | tail -18 | tee -a gpurun_out/sweep_r04n.txt
;;
r04o)
# padding of 56 KB: a GEMM (72 KB), a pooling (56 KB) and an alpha (29 KB) workgroup fit one CU together; with / without wave priorities
for rep in 1 2; do
tools/env_sweep.sh "INFV_NONE=0" "INFV_PR_PAD=57344" "INFV_PR_PAD=57344 INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2" "INFV_PR_PAD=57344 INFV_POOL_PRIO=1 INFV_UC_PRIO=1 INFV_ALPHA_PRIO=1" "INFV_PR_PAD=57344 INFV_PR_U=4 INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2"
done 2>&1 | tee gpurun_out/sweep_r04o.txt
INFV_PR_PAD=57344 INFV_POOL_PRIO=1 INFV_UC_PRIO=2 INFV_ALPHA_PRIO=2 INFV_WG_STAMPS=1 python tools/residency.py pad56prio 2>&1 | grep -v amdgpu.ids | tail -18 | tee -a gpurun_out/sweep_r04o.txt
;;
r04p)
# CU masks: the first K CUs reserved for role S (its own masked stream), the worker streams masked to the rest
{
INFV_CU_MASK=56 INFV_WG_STAMPS=1 timeout 300 python tools/residency.py mask56 2>&1 | grep -v amdgpu.ids | tail -18
python - <<'PY'
import numpy as np
st=np.load("gpurun_out/wg_stamps_mask56.npy"); st=st[(st[:,1]>0)&(st[:,0]>0)]
hw=st[:,2]&0xffffffff; xcc=st[:,2]>>32; cu=(xcc<<8)|(((hw>>13)&7)<<5)|((hw>>8)&15); k=st[:,3]
rs=set(cu[k==4]); others=set(cu[(k!=4)])
print("CUs used by role S:", len(rs), " by the other kernels:", len(others), " shared:", len(rs&others))
PY
tools/env_sweep.sh "INFV_NONE=0" "INFV_CU_MASK=48" "INFV_CU_MASK=56" "INFV_CU_MASK=56 INFV_PR_PAD=57344" "INFV_CU_MASK=56 INFV_PR_PAD=40960" "INFV_CU_MASK=64 INFV_PR_PAD=57344" "INFV_NONE=1"
} 2>&1 | tee gpurun_out/sweep_r04p.txt
;;
r04q)
# five rotating workspace sets (projection GEMM up to five sub-batches ahead of the UC kernel) with and without CU masks
{
python -m pytest tests/test_timed_path_gpu.py -x -q -k "odd_call or oracle or equals" 2>&1 | tail -2
tools/env_sweep.sh "INFV_NONE=0" "INFV_CU_MASK=64" "INFV_CU_MASK=64 INFV_PR_PAD=57344" "INFV_CU_MASK=64 INFV_PR_PAD=40960" "INFV_CU_MASK=60 INFV_PR_PAD=57344" "INFV_CU_MASK=72 INFV_PR_PAD=57344" "INFV_PR_PAD=57344" "INFV_NONE=1"
} 2>&1 | tee gpurun_out/sweep_r04q.txt
;;
r04r)
{
INFV_CU_MASK=64 INFV_PR_PAD=57344 INFV_WG_STAMPS=1 python tools/residency.py mask64 2>&1 | grep -v amdgpu.ids | tail -18
python tools/launch_table.py gpurun_out/wg_stamps_mask64.npy 20 8
} 2>&1 | tee gpurun_out/sweep_r04r.txt
;;
r04s)
# three against five rotating workspace sets (compile-time INFV_PSETS), same box, alternating
for rep in 1 2 3; do
for n in 3 5; do
INFV_LTM_LIBRARY=$PWD/tools/ab/lib_psets$n.so tools/env_sweep.sh "INFV_PSETS_BUILD=$n"
done; done 2>&1 | tee gpurun_out/sweep_r04s.txt
;;
r04t)
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
;;
r04u)
# fewer, longer pooling workgroups: 2 / 3 / 4 rows per workgroup (grid-stride), default padding
for rep in 1 2; do
tools/env_sweep.sh "INFV_NONE=0" "INFV_PR_WGS=1344" "INFV_PR_WGS=896" "INFV_PR_WGS=672"
done 2>&1 | tee gpurun_out/sweep_r04u.txt
;;
r04v)
# (the "logical" library tools/ab/lib_uc_old.so was built from the previous commit's ltm_uc.hip; the parity-order variant was not kept)
# UC kernel with even boxes first in LDS (parity row order) against the logical row order, same box alternating
{
python -m pytest tests/test_ltm_gpu.py tests/test_timed_path_gpu.py -x -q 2>&1 | tail -2
for rep in 1 2 3; do
INFV_LTM_LIBRARY=$PWD/tools/ab/lib_uc_old.so tools/env_sweep.sh "UC_ROWS=logical"
INFV_LTM_LIBRARY=exp tools/env_sweep.sh "UC_ROWS=parity"
done
} 2>&1 | tee gpurun_out/sweep_r04v.txt
;;
r04x)
# what of a pooling workgroup suffers beside a GEMM workgroup: its adds or its loads?  pooling + GEMM only; the second library's
# pooling kernel keeps its loads and drops its adds (-DINFV_POOL_NOADD, timing only)
export INFV_WG_STAMPS=1 INFV_SKIP=12
{
echo "== default pooling kernel"; INFV_LTM_LIBRARY=exp python tools/residency.py add 2>&1 | grep -E "pool |gemm  |    gemm|    pool"
echo "== no adds"; INFV_LTM_LIBRARY=$PWD/tools/ab/lib_noadd.so python tools/residency.py noadd 2>&1 | grep -E "pool |gemm  |    gemm|    pool"
echo "== no adds, alone (INFV_SKIP=14)"; INFV_SKIP=14 INFV_LTM_LIBRARY=$PWD/tools/ab/lib_noadd.so python tools/residency.py noadd_alone 2>&1 | grep -E "pool "
echo "== default, alone (INFV_SKIP=14)"; INFV_SKIP=14 INFV_LTM_LIBRARY=exp python tools/residency.py add_alone 2>&1 | grep -E "pool "
} 2>&1 | tee gpurun_out/sweep_r04x.txt
;;
r04y)
# pooling kernel with LDS-DMA loads (INFV_POOL_DMA=1): bit-identity, lifetime beside a GEMM workgroup, wall clock in situ
{
python -m pytest tests/test_timed_path_gpu.py -x -q -k kept_variants > gpurun_out/variants_r04y.log 2>&1; grep -E "passed|failed|Error|error" gpurun_out/variants_r04y.log | tail -3
echo "== pooling + GEMM only, register loads"; INFV_WG_STAMPS=1 INFV_SKIP=12 python tools/residency.py a 2>&1 | grep -E "pool |gemm  "
echo "== pooling + GEMM only, LDS-DMA loads"; INFV_POOL_DMA=1 INFV_WG_STAMPS=1 INFV_SKIP=12 python tools/residency.py b 2>&1 | grep -E "pool |gemm  "
echo "== alone, LDS-DMA loads"; INFV_POOL_DMA=1 INFV_WG_STAMPS=1 INFV_SKIP=14 python tools/residency.py c 2>&1 | grep -E "pool "
for rep in 1 2; do tools/env_sweep.sh "INFV_NONE=0" "INFV_POOL_DMA=1"; done
} 2>&1 | tee gpurun_out/sweep_r04y.txt
;;
r04z)
# one 256-chunk shard (what every rank of an 8-GPU run does): sub-batch size, median of 15 calls incl. sync, no collective
for sb in 32 16 24 42 32 20 28; do
INFV_SUB_BATCH=$sb python - <<PY 2>/dev/null | tail -1
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from infinite_video_amd import synth
from infinite_video_amd.engine import LTMEngine
from infinite_video_amd.video_memory import consolidate_video
T, P, D, N, H, DH, Q, L, TAU = 256, 32, 768, 256, 12, 64, 32, 2, 0.75
dev = torch.device("cuda:0")
eng = LTMEngine(N, H, DH, D, P, tau=TAU, sticky=True, n_layers=L, max_q=Q, device=dev, max_batch_chunks=42)
projs = [tuple(torch.from_numpy(a).to(dev) for a in synth.layer_projections(l, D, H * DH)) for l in range(L)]
q = torch.from_numpy(np.stack([synth.layer_query(l, Q, H * DH) for l in range(L)])).to(dev)
u = torch.from_numpy(synth.gibbs_uniforms(256, L)).to(dev)
k = torch.randn(256, T * P, D, device=dev)
for _ in range(3): consolidate_video(eng, k, q, projs, u)
torch.cuda.synchronize()
ts = []
for _ in range(15):
    t1 = time.perf_counter(); consolidate_video(eng, k, q, projs, u); torch.cuda.synchronize(); ts.append(time.perf_counter() - t1)
print("sub-batch $sb: shard256 median %.3f ms  min %.3f ms" % (1e3 * sorted(ts)[7], 1e3 * min(ts)))
PY
done 2>&1 | tee gpurun_out/sweep_r04z.txt
;;
r05a)
# (record of how sweep_r05a/d.txt were produced: at that time the bf16x6 GEMM was the default, INFV_PROJ_FP32=1 selected the fp32-MFMA GEMM and
#  INFV_X6_PIPE chose between the two bf16x6 kernels; now INFV_PROJ_X6=1 opts in and only the single-tile kernel is kept)
# projection GEMM as six bf16 MFMA products of exact three-piece splits (default) against the fp32-MFMA GEMM (INFV_PROJ_FP32=1)
{
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_r05a.log 2>&1; grep -E "passed|failed" gpurun_out/gpu_tests_r05a.log | tail -2
for rep in 1 2 3; do
tools/env_sweep.sh "INFV_PROJ_FP32=1" "INFV_PROJ_FP32=0"
done
INFV_LTM_LIBRARY=exp INFV_WG_STAMPS=1 python tools/residency.py x6 2>&1 | grep -v amdgpu.ids | tail -18
python tools/launch_table.py gpurun_out/wg_stamps_x6.npy 20 6
} 2>&1 | tee gpurun_out/sweep_r05a.txt
;;
r05b)
# with the bf16x6 projection GEMM: who sits where, and is the pooling's workgroup rate (9 per us) a dispatch limit?
{
INFV_WG_STAMPS=1 python tools/residency.py x6 2>&1 | grep -v amdgpu.ids | tail -20
python tools/launch_table.py gpurun_out/wg_stamps_x6.npy 20 5
tools/env_sweep.sh "INFV_NONE=0" "INFV_PR_WGS=1344" "INFV_PR_PAD=57344" "INFV_PR_U=4" "INFV_PR_U=4 INFV_PR_PAD=57344" "INFV_NONE=1" "INFV_PR_WGS=1344 INFV_PR_PAD=57344"
} 2>&1 | tee gpurun_out/sweep_r05b.txt
;;
r05c)
# bf16x6 GEMM held to one workgroup per CU by its LDS size (81 KB), pooling padding 78.5 KB so that one of each fits a CU
{
tools/env_sweep.sh "INFV_NONE=0" "INFV_X6_LDS=82944 INFV_PR_PAD=80384" "INFV_X6_LDS=82944 INFV_PR_PAD=80384 INFV_PR_U=4" "INFV_NONE=1" "INFV_X6_LDS=82944 INFV_PR_PAD=80384"
INFV_X6_LDS=82944 INFV_PR_PAD=80384 INFV_WG_STAMPS=1 python tools/residency.py x6l 2>&1 | grep -v amdgpu.ids | tail -20
python tools/launch_table.py gpurun_out/wg_stamps_x6l.npy 20 5
} 2>&1 | tee gpurun_out/sweep_r05c.txt
;;
r05d)
# (record of how sweep_r05a/d.txt were produced: at that time the bf16x6 GEMM was the default, INFV_PROJ_FP32=1 selected the fp32-MFMA GEMM and
#  INFV_X6_PIPE chose between the two bf16x6 kernels; now INFV_PROJ_X6=1 opts in and only the single-tile kernel is kept)
# pipelined bf16x6 projection GEMM (default) against the single-tile one (INFV_X6_PIPE=0) and the fp32-MFMA GEMM (INFV_PROJ_FP32=1)
{
python -m pytest tests/test_ltm_gpu.py -x -q -k "bf16x6" 2>&1 | grep -E "passed|failed" | tail -1
python -m pytest tests/test_ltm_gpu.py tests/test_timed_path_gpu.py -x -q --deselect tests/test_timed_path_gpu.py::test_kept_variants_reproduce_the_default_bit_for_bit > gpurun_out/gpu_tests_r05d.log 2>&1; grep -E "passed|failed" gpurun_out/gpu_tests_r05d.log | tail -1
for rep in 1 2; do
tools/env_sweep.sh "INFV_PROJ_FP32=1" "INFV_X6_PIPE=0" "INFV_NONE=0"
done
INFV_WG_STAMPS=1 python tools/residency.py x6p 2>&1 | grep -v amdgpu.ids | tail -20
python tools/launch_table.py gpurun_out/wg_stamps_x6p.npy 20 5
} 2>&1 | tee gpurun_out/sweep_r05d.txt
;;
r05e)
# (record: INFV_X6_PIPE selected a pipelined bf16x6 kernel that was not kept)
# bf16x6 projection GEMM as a short exclusive burst: the pipelined kernel with six tiles in flight (244 registers, 72 KB: two
# workgroups per CU, no room for a pooling workgroup beside them)
{
for rep in 1 2; do
tools/env_sweep.sh "INFV_X6_PIPE=0" "INFV_X6_PIPE=1"
done
INFV_X6_PIPE=1 INFV_WG_STAMPS=1 python tools/residency.py x6b 2>&1 | grep -v amdgpu.ids | tail -20
python tools/launch_table.py gpurun_out/wg_stamps_x6b.npy 20 5
} 2>&1 | tee gpurun_out/sweep_r05e.txt
;;
r05f)
# more pooling seats (padding 44 / 56 KB, 40-register instantiation) with fewer role-S launches (84- / 126-chunk sub-batches)
run() {  # $1 = batch chunks, rest = env
  bc=$1; shift
  out=$(env "$@" python bench.py --steps 6 --warmup 2 --batch-chunks $bc --no-cpu-baseline --no-encode-video --no-selfcheck --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_pass']
print(round(d['value']), 'chunks/s wall', round(d['ms_per_step'],2), 'pool', k['pool'], 'project', k['project'], 'chain', k['chain'], 'uc', k['uc'])")
  echo "sweep [batch $bc $*] $out"
}
{
run 42 INFV_NONE=0
run 84 INFV_PR_PAD=57344
run 84 INFV_PR_PAD=45056 INFV_PR_U=4
run 126 INFV_PR_PAD=57344
run 126 INFV_PR_PAD=45056 INFV_PR_U=4
run 168 INFV_PR_PAD=45056 INFV_PR_U=4
run 84 INFV_NONE=0
} 2>&1 | tee gpurun_out/sweep_r05f.txt
;;
r05g)
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
;;
r05h)
# single wave-priority changes again, clean experiments build: alpha kernel, UC kernel, role S without its priority
for rep in 1 2; do
tools/env_sweep.sh "INFV_NONE=0" "INFV_ALPHA_PRIO=1" "INFV_UC_PRIO=1" "INFV_ALPHA_PRIO=1 INFV_UC_PRIO=1" "INFV_S_FLAGS=1"
done 2>&1 | tee gpurun_out/sweep_r05h.txt
;;
r05i)
# sub-batch size of long calls, shipped library, same box alternating
run() {
  out=$(python bench.py --steps 8 --warmup 2 --batch-chunks $1 --no-cpu-baseline --no-encode-video --no-selfcheck --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernel_ms_per_pass']
print(round(d['value']), 'chunks/s wall', round(d['ms_per_step'],2), 'pool', k['pool'], 'project', k['project'], 'chain', k['chain'], 'uc', k['uc'])")
  echo "sweep [batch $1] $out"
}
for rep in 1 2 3; do run 42; run 64; run 84; done 2>&1 | tee gpurun_out/sweep_r05i.txt
;;
list) echo r03a r03b r03c r03d r03e r03f r03g r03h r03i r03k r03l r03m r03n r03o r03p r03q r03r r03s r03t r03u r03v r03w r03x r03y r03z r04a r04b r04c r04d r04e r04f r04g r04h r04i r04j r04k r04l r04n r04o r04p r04q r04r r04s r04t r04u r04v r04x r04y r04z r05a r05b r05c r05d r05e r05f r05g r05h r05i ;;
*) echo "usage: tools/sweeps.sh <name>   (tools/sweeps.sh list)"; exit 2 ;;
esac
