export INFV_LTM_LIBRARY=exp INFV_WG_STAMPS=1
python tools/residency.py shard_base 256 > /dev/null 2>&1
python tools/shard_timeline.py gpurun_out/wg_stamps_shard_base.npy 2048
echo =========
INFV_TAPER=16,36,16 INFV_SMALL_TILES=7 python tools/residency.py shard_taper 256 > /dev/null 2>&1
python tools/shard_timeline.py gpurun_out/wg_stamps_shard_taper.npy 1024
