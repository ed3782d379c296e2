#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}   # experiment knobs passed in by the caller only exist in the experiments build (csrc/knobs.h)
# usage (GPU box): tools/trace_window.sh <tag> [chunks]   -- kernel timeline of the last pass: per-stream busy time,
# per-kernel totals, and a window of three sub-batches from the middle (start / end / duration per kernel and stream)
tag=$1; chunks=${2:-2048}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
d=gpurun_out/trace_$tag; rm -rf $d; mkdir -p $d
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 tools/one_pass.py $chunks 3 > $d/run.log 2> $d/err.log
tail -3 $d/run.log
python3 - <<PY
import csv, glob, collections
f = glob.glob("$d/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"]) for r in csv.DictReader(open(f))]
rows = sorted(r for r in rows if "infv::" in r[2])
firsts = [r[0] for r in rows if "qtilde_kernel" in r[2]]
t0 = firsts[-1]
sel = [r for r in rows if r[0] >= t0]
t_end = max(r[1] for r in sel)
print("last pass wall us", (t_end - t0) / 1000)
def short(n):
    n = n.replace("infv::", "").split("(")[0]
    return n[:34]
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n, st in sel:
    agg[(st, short(n))][0] += 1; agg[(st, short(n))][1] += e - s
for (st, kname), (n, t) in sorted(agg.items()):
    print(f"stream {st} {kname:36s} n={n:5d} total {t/1000:9.1f} us avg {t/n/1000:8.2f} us")
by = collections.defaultdict(list)
for s, e, n, st in sel: by[st].append((s, e))
for st, iv in sorted(by.items()):
    iv.sort(); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    tot += ce - cs
    print("stream", st, "busy us", tot / 1000)
chains = [r for r in sel if "chain_batch" in r[2]]
if len(chains) > 6:
    w0 = chains[len(chains) // 2][0]; w1 = chains[len(chains) // 2 + 3][0]
    print("--- window of three sub-batches (us from the start of a chain launch) ---")
    for s, e, n, st in sel:
        if e >= w0 and s <= w1:
            print(f"{(s-w0)/1000:9.1f} {(e-w0)/1000:9.1f} dur {(e-s)/1000:8.1f}  st {st:>3} {short(n)}")
PY
