#!/bin/bash
# one 256-chunk shard (what every rank of an 8-GPU run does): sub-batch size, median of 15 calls incl. sync, no collective
export INFV_LTM_LIBRARY=exp
for sb in 32 16 24 42 32 20 28; do
INFV_SUB_BATCH=$sb python - <<PY 2>/dev/null | tail -1
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from infinite_video_amd import synth
from infinite_video_amd.engine import LTMEngine
from infinite_video_amd.video_memory import consolidate_video
T, P, D, N, H, DH, Q, L, TAU = 256, 32, 768, 256, 12, 64, 32, 2, 0.75
dev = torch.device("cuda:0")
eng = LTMEngine(N, H, DH, D, P, tau=TAU, sticky=True, n_layers=L, max_q=Q, device=dev, max_batch_chunks=42)
projs = [tuple(torch.from_numpy(a).to(dev) for a in synth.layer_projections(l, D, H * DH)) for l in range(L)]
q = torch.from_numpy(np.stack([synth.layer_query(l, Q, H * DH) for l in range(L)])).to(dev)
u = torch.from_numpy(synth.gibbs_uniforms(256, L)).to(dev)
k = torch.randn(256, T * P, D, device=dev)
for _ in range(3): consolidate_video(eng, k, q, projs, u)
torch.cuda.synchronize()
ts = []
for _ in range(15):
    t1 = time.perf_counter(); consolidate_video(eng, k, q, projs, u); torch.cuda.synchronize(); ts.append(time.perf_counter() - t1)
print("sub-batch $sb: shard256 median %.3f ms  min %.3f ms" % (1e3 * sorted(ts)[7], 1e3 * min(ts)))
PY
done 2>&1 | tee gpurun_out/sweep_r04z.txt
