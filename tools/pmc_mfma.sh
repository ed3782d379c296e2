#!/bin/bash
# MFMA / LDS counters per kernel (rocprofv3 --pmc, one pass per counter group, kernel-trace only):
#   ./tools/pmc_mfma.sh <tag> -- python3 <script> [args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift; shift
# The profiler's preloaded library initialises the GPU before the program starts, and a process that has touched the GPU
# must not exec another program on this pool (it takes the machine down).  So the command after `--` must BE the
# program: python3 / python or a binary -- never env, taskset, numactl, bash -c, or a #! script that re-execs.
# Export knobs in the calling shell (export VAR=...; ./tools/pmc_mfma.sh ...) instead of prefixing them with `env`.
case "$(basename -- "$1")" in
  python3|python|python3.*) ;;
  env|taskset|numactl|bash|sh|time|timeout|nice|*.py|*.sh)
    echo "pmc_mfma.sh: refusing to profile through '$1' (an exec hop after GPU initialisation); run python3 <script> directly" >&2; exit 2;;
  *) if head -c 2 -- "$1" 2>/dev/null | grep -q '^#!'; then echo "pmc_mfma.sh: '$1' is a #! script; run its interpreter directly" >&2; exit 2; fi;;
esac
out=gpurun_out/pmc_$tag; rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/mfma -- "$@" > $out/mfma.out 2> $out/mfma.err
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $out/lds -- "$@" > $out/lds.out 2> $out/lds.err
python3 - <<PY
import csv, glob, collections, json
out = "$out"
res = {}
for grp in ("mfma", "lds"):
    fs = glob.glob(out + f"/{grp}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print("no file for", grp); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        res.setdefault(k, {}).update({n: v for n, v in d.items()})
        res[k]["dispatches"] = max(cnt[(k, n)] for n in d)
rows = []
for k, d in res.items():
    if "GRBM_GUI_ACTIVE" in d and d["GRBM_GUI_ACTIVE"] > 0:
        # SQ_VALU_MFMA_BUSY_CYCLES is summed over SIMDs (4 per CU, 256 CUs); GRBM_GUI_ACTIVE is the dispatch's active
        # cycles summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS): wall cycles = GRBM / 8
        d["mfma_util_pct"] = 100.0 * d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (d["GRBM_GUI_ACTIVE"] / 8.0 * 4 * 256)
    if d.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        d["lds_bank_conflict_pct"] = 100.0 * d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
    rows.append((d.get("GRBM_GUI_ACTIVE", 0.0), k, d))
rows.sort(reverse=True)
json.dump({k: d for _, k, d in rows}, open(out + "/pmc_mfma_summary.json", "w"), indent=1)
for _, k, d in rows[:12]:
    print(f"{k:60s} n={int(d.get('dispatches', 0)):4d} mfma_util={d.get('mfma_util_pct', float('nan')):6.1f}%  lds_conflict={d.get('lds_bank_conflict_pct', float('nan')):5.1f}%")
PY
