#!/usr/bin/env python3
"""Race screen: two engines (a 2048-chunk video and a 256-chunk shard) called alternately many times on the shared worker
streams; every result must be bit-identical to the first one of its kind (the fixed-point histogram makes the path
order-independent, so any difference is a race or a stale buffer)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from infinite_video_amd import synth
from infinite_video_amd.engine import LTMEngine
from infinite_video_amd.video_memory import consolidate_video

T, P, D, N, H, DH, Q, L, TAU = 256, 32, 768, 256, 12, 64, 32, 2, 0.75
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
big = LTMEngine(N, H, DH, D, P, tau=TAU, sticky=True, n_layers=L, max_q=Q, device=dev, max_batch_chunks=42)
small = LTMEngine(N, H, DH, D, P, tau=TAU, sticky=True, n_layers=L, max_q=Q, device=dev, max_batch_chunks=42)
projs = [tuple(torch.from_numpy(a).to(dev) for a in synth.layer_projections(l, D, H * DH)) for l in range(L)]
q = torch.from_numpy(np.stack([synth.layer_query(l, Q, H * DH) for l in range(L)])).to(dev)
q2 = torch.from_numpy(np.stack([synth.layer_query(7 + l, Q, H * DH) for l in range(L)])).to(dev)
u = torch.from_numpy(synth.gibbs_uniforms(2048, L)).to(dev)
k = torch.empty(2048, T * P, D, device=dev)
gen = torch.Generator(device=dev).manual_seed(3)
for i in range(0, 2048, 64):
    k[i:i + 64].normal_(generator=gen)
torch.cuda.synchronize()
ref_big = ref_small = None
bad = 0
t0 = time.perf_counter()
for it in range(iters):
    a, ma = consolidate_video(big, k, q, projs, u)
    b, mb = consolidate_video(small, k[512:768], q2, projs, u[512:768])
    if ref_big is None:
        ref_big, ref_small = (a.clone(), ma.B.clone()), (b.clone(), mb.B.clone())
    else:
        ok = torch.equal(a, ref_big[0]) and torch.equal(ma.B, ref_big[1]) and torch.equal(b, ref_small[0]) and torch.equal(mb.B, ref_small[1])
        bad += 0 if ok else 1
torch.cuda.synchronize()
print(f"soak: {iters} iterations, {bad} mismatching, {1e3 * (time.perf_counter() - t0) / iters:.2f} ms per iteration")
sys.exit(1 if bad else 0)
