#!/bin/bash
# split_gemm_wide_kernel in the layer-major Q-former path: per-launch durations (kernel trace) and MFMA-busy (PMC pass)
# usage (GPU box): bash tools/r04_wide.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r04}
rm -rf gpurun_out/${tag}_qf_trace
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_qf_trace -- python3 tools/bench_qformer.py --batched 64 --calls 3 > gpurun_out/${tag}_qf_trace.json 2>/dev/null
python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/${tag}_qf_trace/**/*kernel_trace.csv",recursive=True)[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "split_gemm_wide" in r["Kernel_Name"]:
        agg[(r["Grid_Size_Y"],r["Grid_Size_Z"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(agg.items()):
    v=sorted(v); print("split_gemm_wide grid.y/256, grid.z",k,"n",len(v),"median us",v[len(v)//2],"min",v[0])
f=glob.glob("gpurun_out/${tag}_qf_trace/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.reader(open(f)))[:8]: print(r[0][:70], r[1:5])
PY
bash tools/pmc_mfma.sh ${tag}_qformer -- python3 tools/bench_qformer.py --batched 64 --calls 2 2>&1 | grep -i "split_gemm\|qf_gemm"
