#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # experiment knobs passed in by the caller only exist in the experiments build (csrc/knobs.h)
# per-kernel time of the layer-major video Q-former path (252 chunks).  usage (GPU box): tools/trace_qformer.sh <tag> [env...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-x}; shift
for kv in "$@"; do export "$kv"; done
out=gpurun_out/qft_$tag; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/bench_qformer.py --batched 252 --calls 3 > $out/bench.json 2> $out/err.txt
python3 - <<PY
import csv,glob
f=glob.glob("$out/trace/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.reader(open(f)))
tot=0
for r in rows[1:14]:
    print(r[0][:50].ljust(50), r[1].rjust(6), "per chunk-pass us", round(float(r[2])/1008/1e3,2), "avg us", round(float(r[3])/1e3,1))
print(open("$out/bench.json").read()[100:200])
PY
