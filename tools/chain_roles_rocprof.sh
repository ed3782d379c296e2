#!/bin/bash
# kernel-trace of the chain kernel with role subsets (timing experiments only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in ${ROLES:-7 1 2 4}; do
  export INFV_CHAIN_ROLES=$m
  d=gpurun_out/roles_$m; mkdir -p $d
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --steps 1 --warmup 1 --chunks 256 --no-cpu-baseline > $d/bench.json 2> $d/err.log
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "roles=$m"; grep -E "chain_kernel|new_scores|pool_frames|gemm_nt" $f | cut -c1-60,200- | sed 's/.*\(chain_kernel\|new_scores_kernel\|pool_frames_kernel\|gemm_nt_kernel<[0-9, ]*>\).*)",/\1,/' 
done
