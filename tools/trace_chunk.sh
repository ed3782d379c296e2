#!/bin/bash
# per-chunk-call video Q-former path (encode_video counterpart): kernel us per chunk by kernel, their sum against the wall per chunk
# usage (GPU box): tools/trace_chunk.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-x}
out=gpurun_out/qfc_$tag; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/bench_qformer.py --chunks 64 > $out/bench.json 2> $out/err.txt
python3 - <<PY
import csv,glob,json
f=glob.glob("$out/trace/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.reader(open(f)))
n=68.0          # 4 warm-up + 64 timed chunk calls
tot=0; launches=0
for r in rows[1:]:
    if "at::native" in r[0] or "rocclr" in r[0]: continue
    tot+=float(r[2]); launches+=int(r[1])
for r in rows[1:22]:
    print(r[0][:64].ljust(64), "calls/chunk", round(int(r[1])/n,1), "us/chunk", round(float(r[2])/n/1e3,1), "avg us", round(float(r[3])/1e3,1))
print("library kernels: launches per chunk", round(launches/n,1), " kernel us per chunk (sum)", round(tot/n/1e3,1))
print(open("$out/bench.json").read().strip().split("\n")[-1])
PY
