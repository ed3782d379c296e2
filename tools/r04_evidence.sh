#!/bin/bash
# the round's evidence passes in one call: profile_round.sh <tag> (kernel trace + FETCH/WRITE PMC passes of the headline bench) and
# pmc_mfma.sh on a 512-chunk consolidation.  usage (GPU box): bash tools/r04_evidence.sh <tag>
tag=${1:-r04_b}
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh $tag > gpurun_out/prof_$tag.log 2>&1
tail -15 gpurun_out/prof_$tag.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/pmc_mfma.sh ${tag}_ltm -- python3 tools/one_pass.py 512 2 2>&1 | tail -12
