export INFV_LTM_LIBRARY=exp
for r in 1 2; do
  echo "== DMA=0"; INFV_CHAIN_DMA=0 tools/quick_bench.sh a_$r 6 2>&1 | tail -1
  echo "== DMA=1 U=4"; INFV_PR_U=4 tools/quick_bench.sh d_$r 6 2>&1 | tail -1
  echo "== DMA=1 U=2"; INFV_PR_U=2 tools/quick_bench.sh e_$r 6 2>&1 | tail -1
  echo "== DMA=0 U=4"; INFV_CHAIN_DMA=0 INFV_PR_U=4 tools/quick_bench.sh f_$r 6 2>&1 | tail -1
done
