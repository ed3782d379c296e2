#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
d=gpurun_out/overlap; rm -rf $d; mkdir -p $d
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 bench.py --steps 1 --warmup 1 --chunks 256 --no-cpu-baseline > $d/bench.json 2> $d/err.log
python3 - <<PY
import csv,glob
f=glob.glob("$d/**/*kernel_trace.csv",recursive=True)[0]
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"][:30],r["Stream_Id"] if "Stream_Id" in r else r.get("Queue_Id")) for r in csv.DictReader(open(f))]
rows.sort()
# take a window in the middle of the run
mid=[r for r in rows if "chain_kernel" in r[2] or "pool_frames" in r[2] or "gemm_nt" in r[2] or "new_scores" in r[2] or "build_rows" in r[2]]
n=len(mid)
t0=mid[n//2][0]
for s,e,name,st in mid[n//2:n//2+60]:
    print(f"{(s-t0)/1000:9.1f} {(e-t0)/1000:9.1f} dur {(e-s)/1000:7.1f} us  stream {st} {name}")
PY
