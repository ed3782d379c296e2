#!/bin/bash
# Round-6 evidence in one GPU call: (1) rocprofv3 kernel stats + FETCH/WRITE PMC passes of the headline bench (profile_round.sh),
# (2) MFMA-busy / LDS-conflict PMC passes on a 512-chunk consolidation, (3) residency + empty-CU accounting of the shipped pipeline
# and of role S with the LDS-DMA loader sharing its CUs (INFV_CHAIN_DMA=1), (4) the full default bench line.
# Copy the summaries to profiles/r06_*.
export PYTHONUNBUFFERED=1
tools/profile_round.sh r06 > gpurun_out/r06_profile_round.txt 2>&1
( export INFV_LTM_LIBRARY= ; tools/pmc_mfma.sh r06_ltm -- python3 tools/one_pass.py 512 2 > gpurun_out/r06_pmc_mfma.txt 2>&1 )
{
res() { echo "== $1"; local tag=$2; shift; shift; env INFV_LTM_LIBRARY=exp INFV_WG_STAMPS=1 "$@" timeout 300 python tools/residency.py $tag 2>&1 | grep -v amdgpu.ids | tail -13;
        python tools/empty_cu.py gpurun_out/wg_stamps_$tag.npy 2>&1 | sed 's/^/   /'; python tools/role_s_gaps.py gpurun_out/wg_stamps_$tag.npy | sed 's/^/   /'; }
res "shipped pipeline (ONE pooling launch per call; role S -- register loader, a CU each --, GEMM, alpha and UC per sub-batch)" r06
res "round 5's form: one pooling launch per sub-batch, the caller's stream also waits for the UC kernel of five sub-batches ago" r06old INFV_POOL_CALL=0 INFV_DROP_WAITS=0
res "role S with the LDS-DMA loader (128 registers), unpadded: pooling workgroups share its CUs (round 5's launch form)" r06dma INFV_CHAIN_DMA=1 INFV_POOL_CALL=0 INFV_DROP_WAITS=0
res "role S with the LDS-DMA loader, 77 KB of LDS: two of them per CU, no pooling workgroup beside them (round 5's launch form)" r06dma77 INFV_CHAIN_DMA=1 INFV_S_LDS=78848 INFV_POOL_CALL=0 INFV_DROP_WAITS=0
res "round 5's form without the UC and alpha launches (INFV_SKIP=4)" r06skip4 INFV_SKIP=4 INFV_POOL_CALL=0 INFV_DROP_WAITS=0
} > gpurun_out/r06_residency.txt 2>&1
python bench.py > gpurun_out/r06_a_bench.json 2> gpurun_out/r06_a_bench.err
tail -c 600 gpurun_out/r06_a_bench.json
