export INFV_LTM_LIBRARY=exp
B=$PWD/infinite-video_amd/libinfv_ltm_v_r05base.so
for r in 1 2 3; do
  echo "== r05 base lib"; INFV_LTM_LIBRARY=$B tools/quick_bench.sh z_$r 6 2>&1 | tail -1
  echo "== DMA=0"; INFV_CHAIN_DMA=0 tools/quick_bench.sh a_$r 6 2>&1 | tail -1
  echo "== DMA=1 S_LDS=83968"; INFV_S_LDS=83968 tools/quick_bench.sh c_$r 6 2>&1 | tail -1
done
