#!/bin/bash
# kernel-trace stats of the drop-in module's per-call path (tools/bench_module.py): us per kernel per forward
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
d=gpurun_out/trace_module; rm -rf $d; mkdir -p $d
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_module.py > $d/run.log 2> $d/err.log
tail -2 $d/run.log
python3 - <<PY
import csv, glob
f = glob.glob("$d/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6} avg {float(r['AverageNs'])/1000:8.2f} us")
PY
