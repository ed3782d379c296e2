#!/bin/bash
# per-stream busy time and kernel summary of one bench pass (256 chunks)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
d=gpurun_out/timeline; rm -rf $d; mkdir -p $d
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 bench.py --steps 1 --warmup 1 --chunks ${CHUNKS:-256} --no-cpu-baseline > $d/bench.json 2> $d/err.log
python3 - <<PY
import csv,glob,collections
f=glob.glob("$d/**/*kernel_trace.csv",recursive=True)[0]
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"],r["Stream_Id"]) for r in csv.DictReader(open(f))]
rows=[r for r in rows if "infv::" in r[2]]
rows.sort()
# last pass only: take the final 45% of kernels by time
t_end=max(r[1] for r in rows)
# find start of last consolidate: last first-chunk gemm<64,64>
starts=[r[0] for r in rows if "pool_frames_kernel<16, 256>" in r[2]]
t0=starts[-1] if starts else rows[0][0]
sel=[r for r in rows if r[0]>=t0]
print("pass wall us", (t_end-t0)/1000)
agg=collections.defaultdict(lambda:[0,0])
for s,e,n,st in sel:
    k=n.split("(")[0][-28:]
    agg[(st,k)][0]+=1; agg[(st,k)][1]+=e-s
for (st,k),(n,t) in sorted(agg.items()):
    print(f"stream {st} {k:30s} n={n:5d} total {t/1000:9.1f} us avg {t/n/1000:7.2f} us")
# busy time per stream (union)
by=collections.defaultdict(list)
for s,e,n,st in sel: by[st].append((s,e))
for st,iv in by.items():
    iv.sort(); tot=0; cs,ce=iv[0]
    for s,e in iv[1:]:
        if s>ce: tot+=ce-cs; cs,ce=s,e
        else: ce=max(ce,e)
    tot+=ce-cs
    print("stream",st,"busy us",tot/1000)
PY
