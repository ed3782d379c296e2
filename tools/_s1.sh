export INFV_LTM_LIBRARY=exp
for r in 1 2 3; do
for m in 0 1 2; do echo -n "POOL_STORE=$m "; INFV_POOL_STORE=$m tools/quick_bench.sh st${m}_$r 6 2>&1 | tail -1; done
done
