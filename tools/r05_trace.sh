#!/bin/bash
export INFV_LTM_LIBRARY=${INFV_LTM_LIBRARY:-exp}
# usage (GPU box): tools/r05_trace.sh <tag> [chunks] [window_start_us] [window_len_us]
# kernel timeline of the last pass of tools/one_pass.py: which HIP stream sits on which hardware queue, per-stream busy time,
# per-kernel totals and a window of the timeline (start / end / duration, stream, queue)
tag=$1; chunks=${2:-512}; w0=${3:-0}; wl=${4:-900}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
d=gpurun_out/trace_$tag; rm -rf $d; mkdir -p $d
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 tools/one_pass.py $chunks 3 > $d/run.log 2> $d/err.log
tail -3 $d/run.log
python3 - <<PY
import csv, glob, collections
f = glob.glob("$d/**/*kernel_trace.csv", recursive=True)[0]
rd = list(csv.DictReader(open(f)))
print("columns:", list(rd[0].keys()))
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"], r.get("Queue_Id", "?")) for r in rd)
firsts = [r[0] for r in rows if "qtilde_kernel" in r[2]]
t0 = firsts[-1]
sel = [r for r in rows if r[0] >= t0]
t_end = max(r[1] for r in sel)
print("last pass wall us", (t_end - t0) / 1000)
print("stream -> queue:", sorted({(r[3], r[4]) for r in sel}))
def short(n):
    return n.replace("infv::", "").replace("void ", "").split("(")[0][:34]
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n, st, q in sel:
    agg[(st, short(n))][0] += 1; agg[(st, short(n))][1] += e - s
for (st, kname), (n, t) in sorted(agg.items()):
    print(f"stream {st} {kname:36s} n={n:5d} total {t/1000:9.1f} us avg {t/n/1000:8.2f} us")
print("--- window ---")
for s, e, n, st, q in sel:
    if (e - t0) / 1000 >= $w0 and (s - t0) / 1000 <= $w0 + $wl:
        print(f"{(s-t0)/1000:9.1f} {(e-t0)/1000:9.1f} dur {(e-s)/1000:8.1f}  st {st:>3} q {q:>3} {short(n)}")
PY
