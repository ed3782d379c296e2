#!/bin/bash
# where the waves of each kernel spend their cycles: parked (s_waitcnt / barrier), issue-stalled, LDS-issue-stalled, issuing
#   ./tools/pmc_wait.sh <tag> -- python3 <script> [args]      (program directly after --: see pmc_mfma.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift; shift
case "$(basename -- "$1")" in python3|python|python3.*) ;; *) echo "pmc_wait.sh: run python3 <script> directly" >&2; exit 2;; esac
out=gpurun_out/pmcw_$tag; rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/sq -- "$@" > $out/out.txt 2> $out/err.txt
python3 - <<PY
import csv, glob, collections
fs = glob.glob("$out/sq/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(fs[0])):
    agg[r["Kernel_Name"][:48]][r["Counter_Name"]] += float(r["Counter_Value"])
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]
for k, d in rows:
    wc = d.get("SQ_WAVE_CYCLES", 1) or 1
    print(f"{k:48s} parked {100*d.get('SQ_WAIT_ANY',0)/wc:5.1f}%  issue-stall {100*d.get('SQ_WAIT_INST_ANY',0)/wc:5.1f}% (LDS {100*d.get('SQ_WAIT_INST_LDS',0)/wc:5.1f}%)  issuing {100*d.get('SQ_ACTIVE_INST_ANY',0)/wc:5.1f}%  mfma-busy/wave-cycles {100*d.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/wc:5.1f}%  lds-conflict {100*d.get('SQ_LDS_BANK_CONFLICT',0)/max(d.get('SQ_LDS_IDX_ACTIVE',1),1):5.1f}%")
PY
