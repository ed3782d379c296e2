L=$PWD/infinite-video_amd/libinfv_ltm_v_dma.so
export INFV_LTM_LIBRARY=$L
for r in 1 2; do
  echo "== DMA=0"; INFV_CHAIN_DMA=0 tools/quick_bench.sh a_$r 6 2>&1 | tail -1
  echo "== DMA=1 S_LDS=78848"; INFV_S_LDS=78848 tools/quick_bench.sh b_$r 6 2>&1 | tail -1
  echo "== DMA=1 S_LDS=83968"; INFV_S_LDS=83968 tools/quick_bench.sh c_$r 6 2>&1 | tail -1
done
echo "== residency S_LDS=78848"; INFV_S_LDS=78848 INFV_WG_STAMPS=1 python tools/residency.py dma77 2048 2>&1 | tail -16
