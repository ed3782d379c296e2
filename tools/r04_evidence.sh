cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r04_b > gpurun_out/prof_r04_b.log 2>&1
tail -15 gpurun_out/prof_r04_b.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/pmc_mfma.sh r04_b_ltm -- python3 tools/one_pass.py 512 2 2>&1 | tail -12
