#!/usr/bin/env python3
"""Two warm-up passes + one pass of the headline consolidation (nothing else): what the trace tools profile."""
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
chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
eng = LTMEngine(N, H, DH, D, P, tau=TAU, sticky=True, n_layers=L, max_q=Q, device=dev, max_batch_chunks=42)
projs = [tuple(torch.from_numpy(a).to(dev) for a in synth.layer_projections(l, D, H * DH)) for l in range(L)]
q = torch.from_numpy(np.stack([synth.layer_query(l, Q, H * DH) for l in range(L)])).to(dev)
u = torch.from_numpy(synth.gibbs_uniforms(chunks, L)).to(dev)
k = torch.empty(chunks, T * P, D, device=dev)
gen = torch.Generator(device=dev).manual_seed(1)
for i in range(0, chunks, 64):
    k[i:i + 64].normal_(generator=gen)
if len(sys.argv) > 3 and sys.argv[3] == "bf16":     # frame tokens as the optional bf16 producer layout (halves the pooled bytes)
    k16 = torch.empty(chunks, T * P, D, device=dev, dtype=torch.bfloat16)
    for i in range(0, chunks, 64):
        k16[i:i + 64] = k[i:i + 64]
    k = k16
torch.cuda.synchronize()
prio = os.environ.get("ONE_PASS_PRIO")                 # run the call on a torch stream of this priority (-1 = most urgent)
stream = torch.cuda.Stream(device=dev, priority=int(prio)) if prio is not None else torch.cuda.current_stream(dev)
for p in range(passes):
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        consolidate_video(eng, k, q, projs, u)
    torch.cuda.synchronize()
    print(f"pass {p}: {1e3 * (time.perf_counter() - t0):.3f} ms", flush=True)
