#!/bin/bash
# per-chunk-call video Q-former path: kernel time per chunk vs wall per chunk (how launch-bound it is)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/qfc_${1:-x}; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/bench_qformer.py --chunks 64 > $out/bench.json 2> $out/err.txt
python3 - <<PY
import csv,glob
f=glob.glob("$out/trace/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.reader(open(f)))
n=68
tot=sum(float(r[2]) for r in rows[1:])/n/1e3
calls=sum(int(r[1]) for r in rows[1:])/n
print("kernel us per chunk", round(tot,1), "launches per chunk", round(calls,1))
for r in rows[1:16]: print(r[0][:50].ljust(50), "calls/chunk", round(int(r[1])/n,1), "us/chunk", round(float(r[2])/n/1e3,2), "avg us", round(float(r[3])/1e3,1))
print(open("$out/bench.json").read()[60:170])
PY
