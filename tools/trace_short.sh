#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # experiment knobs passed in by the caller only exist in the experiments build (csrc/knobs.h)
# usage (GPU box): tools/trace_short.sh <tag> [chunks=256]  -- full kernel timeline of the last pass of a SHORT call
# (the 8-GPU shard of the headline video): where the fill / drain / packing time goes
tag=$1; chunks=${2:-256}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
d=gpurun_out/short_$tag; rm -rf $d; mkdir -p $d
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $d -- python3 tools/one_pass.py $chunks 6 > $d/run.log 2> $d/err.log
tail -3 $d/run.log
python3 - <<PY
import csv, glob
f = glob.glob("$d/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"]) for r in csv.DictReader(open(f))]
rows.sort()
firsts = [r[0] for r in rows if "qtilde_kernel" in r[2]]
t0 = firsts[-1]
prev_end = max(r[1] for r in rows if r[0] < t0)
sel = [r for r in rows if r[0] >= t0]
print("gap before call us", (t0 - prev_end) / 1000, "last pass kernels span us", (max(r[1] for r in sel) - t0) / 1000)
def short(n):
    return n.replace("infv::", "").replace("void ", "").split("(")[0][:30]
for s, e, n, st in sel:
    print(f"{(s-t0)/1000:9.1f} {(e-t0)/1000:9.1f} dur {(e-s)/1000:7.1f} st {st:>3} {short(n)}")
PY
