L=$PWD/infinite-video_amd/libinfv_ltm_v_dma.so
export INFV_LTM_LIBRARY=$L
for r in 1 2; do
for dma in 0 1; do
  echo "== DMA=$dma in situ"; INFV_CHAIN_DMA=$dma tools/quick_bench.sh dma${dma}_$r 6 2>&1 | tail -1
done
done
for dma in 0 1; do
  echo "== DMA=$dma chain only (INFV_SKIP=7)"; INFV_CHAIN_DMA=$dma INFV_SKIP=7 python tools/one_pass.py 2048 5 2>&1 | grep "^pass" | tail -3
  echo "== DMA=$dma chain + pool (INFV_SKIP=6)"; INFV_CHAIN_DMA=$dma INFV_SKIP=6 python tools/one_pass.py 2048 5 2>&1 | grep "^pass" | tail -3
done
