#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # the INFV_* knobs below only exist in the experiments build (csrc/knobs.h)
# FETCH_SIZE of the video Q-former path per chunk, with the single token pass (INFV_VQF_FUSE=1, default) and with the
# round-1 arrangement (separate pooling + one split pass per layer).  One PMC pass each, nothing else traced.
# usage (GPU box): tools/pmc_qformer.sh <tag>  -> gpurun_out/pmcq_<tag>/summary.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r02}
out=gpurun_out/pmcq_$tag; rm -rf $out; mkdir -p $out
for fuse in 1 0; do
  export INFV_VQF_FUSE=$fuse
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/chunk_$fuse -- python3 tools/bench_qformer.py --chunks 16 > $out/chunk_$fuse.json 2> $out/chunk_$fuse.err
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/video_$fuse -- python3 tools/bench_qformer.py --batched 64 --calls 1 > $out/video_$fuse.json 2> $out/video_$fuse.err
done
unset INFV_VQF_FUSE
python3 - <<PY
import csv, glob, collections, json
out = "$out"
res = {}
for mode, chunks in (("chunk", 20), ("video", 128)):        # chunks processed by the profiled program (warm-up included)
    for fuse in (1, 0):
        fs = glob.glob(f"{out}/{mode}_{fuse}/**/*counter_collection.csv", recursive=True)
        if not fs:
            print("no pmc file", mode, fuse); continue
        agg = collections.defaultdict(float)
        for r in csv.DictReader(open(fs[0])):
            agg[r["Kernel_Name"][:70]] += float(r["Counter_Value"])
        # FETCH_SIZE is in KiB and counts 128-B requests as 64 B on gfx950 (MI355X_MICROARCH.md, HBM): x2
        per_chunk = {k: 2.0 * v * 1024 / chunks / 1e6 for k, v in agg.items()}
        top = dict(sorted(per_chunk.items(), key=lambda kv: -kv[1])[:8])
        res[f"{mode}_fuse{fuse}"] = {"chunks": chunks, "fetch_MB_per_chunk": sum(per_chunk.values()), "top_kernels_MB_per_chunk": top}
        print(mode, "fuse", fuse, "fetch MB/chunk", round(sum(per_chunk.values()), 1))
json.dump(res, open(out + "/summary.json", "w"), indent=1)
PY
