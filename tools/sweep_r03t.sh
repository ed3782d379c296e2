#!/bin/bash
# per-CU streaming rate of the pooling kernel alone, against the number of CUs it may use
export INFV_LTM_LIBRARY=exp
{
for w in 32 64 128 192 208 256; do INFV_PR_WGS=$w python tools/pool_cus.py 2>/dev/null | tail -1; done
for w in 256 416 512; do INFV_PR_WGS=$w INFV_PR_PAD=40000 python tools/pool_cus.py 2>/dev/null | tail -1; done
python tools/pool_cus.py 2>/dev/null | tail -1
} | tee gpurun_out/sweep_r03t.txt
