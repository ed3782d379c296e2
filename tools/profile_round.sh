#!/bin/bash
# rocprofv3 passes of the headline bench: kernel-trace stats, then PMC passes (each on its own, as gpurun requires)
# usage (GPU box): tools/profile_round.sh <tag>    -> gpurun_out/prof_<tag>/ (copy the summaries to profiles/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r02}
out=gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-encode-video --no-secondary > $out/bench_trace.json 2> $out/trace.err
# (the kernel-trace pass above and the PMC passes run the SHIPPED library with nothing set.  The PMC passes use a 768-chunk video: the
# shortest call that takes the headline's path -- one pooling launch for the call, 42-chunk sub-batches for everything else; 18 full
# sub-batches + a short one)
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --chunks 768 --no-cpu-baseline --no-encode-video --no-selfcheck --no-secondary > $out/bench_fetch.json 2> $out/fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 1 --warmup 0 --chunks 768 --no-cpu-baseline --no-encode-video --no-selfcheck --no-secondary > $out/bench_write.json 2> $out/write.err
python3 - <<PY
import csv,glob,collections,json
out="$out"
f=glob.glob(out+"/trace/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.reader(open(f)))
w=csv.writer(open(out+"/kernel_stats_summary.csv","w"))
w.writerow(rows[0])
for r in rows[1:]:
    if float(r[4])>0.05: w.writerow([r[0][:110]]+r[1:])
res={}
for name in ("fetch","write"):
    fs=glob.glob(out+f"/pmc_{name}/**/*counter_collection.csv",recursive=True)
    if not fs: print("no pmc file for",name); continue
    agg=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"][:60]+" grid="+r["Grid_Size"]      # per launch size: a short tail launch is not a full one
        agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
    res[name]={k:(n,v/n) for k,(n,v) in agg.items()}
json.dump(res,open(out+"/pmc_summary.json","w"),indent=1)
for name,d in res.items():
    for k,(n,v) in sorted(d.items(), key=lambda kv:-kv[1][1])[:6]:
        print(name,k,n,"avg counter per dispatch",v)
PY
