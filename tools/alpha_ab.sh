#!/bin/bash
# alpha_rows2_kernel per launch in situ, without the pooling stream (INFV_SKIP=1) and without pooling + GEMM + UC (INFV_SKIP=7)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export INFV_LTM_LIBRARY=exp
for skip in 0 1 7; do
export INFV_SKIP=$skip
rm -rf gpurun_out/alpha_$skip
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/alpha_$skip -- python3 tools/one_pass.py 2048 3 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/alpha_$skip/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.reader(open(f)))[1:9]:
    if "at::" in r[0]: continue
    print("skip $skip", r[0][:50].ljust(50), r[1], "avg us", round(float(r[3])/1e3,1))
PY
done
