#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in ${ROLES:-7 1}; do
  export INFV_CHAIN_ROLES=$m
  d=gpurun_out/med_$m; rm -rf $d; mkdir -p $d
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 bench.py --steps 1 --warmup 1 --chunks 256 --no-cpu-baseline > $d/bench.json 2> $d/err.log
  python3 - <<PY
import csv,glob
f=glob.glob("$d/**/*kernel_trace.csv",recursive=True)[0]
v=sorted(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "chain_kernel" in r["Kernel_Name"])
print("roles=$m chain_kernel n=",len(v),"p10",v[len(v)//10],"median",v[len(v)//2],"p90",v[9*len(v)//10])
PY
done
