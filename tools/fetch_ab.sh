#!/bin/bash
# FETCH_SIZE per kernel (KiB per dispatch, x2 for the gfx950 correction) for the previous library and the current one
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for which in prev new; do
  out=gpurun_out/fetch_$which; rm -rf $out; mkdir -p $out
  if [ $which = prev ]; then export INFV_LTM_LIBRARY=$PWD/tools/lib_prev.so; else unset INFV_LTM_LIBRARY; fi
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -- python3 tools/one_pass.py 512 2 > $out/run.log 2> $out/err.log
  python3 - <<PY
import csv,glob,collections
fs=glob.glob("$out/**/*counter_collection.csv",recursive=True)
agg=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open(fs[0])):
    k=r["Kernel_Name"][:50]; agg[k][0]+=1; agg[k][1]+=float(r["Counter_Value"])
for k,(n,v) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:7]:
    print("$which", f"{k:50s} n={n:4d} avg FETCH x2 = {2*v/n/1024:8.1f} MB per dispatch")
PY
done
