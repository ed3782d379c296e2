#!/usr/bin/env python3
"""How much does chunk-block sharding change the result?  (GPU box.)

The reference is a single process: every chunk sees the memory of all earlier chunks.  The multi-GPU mode cuts the video
into R contiguous blocks, each consolidated as its own document (video_memory.py), so chunks after a block boundary start
from an empty memory.  This tool runs both on ONE GPU -- the single stream, and the R blocks one after the other -- on the
headline synthetic video and reports how far the quantities handed to the LLM move: the mean over chunks of the per-chunk
outputs (nextqa.py:194) and the per-chunk outputs themselves."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from infinite_video_amd import synth
from infinite_video_amd.engine import LTMEngine
from infinite_video_amd.video_memory import shard_range

T, P, D, N, H, DH, Q, L, TAU = 256, 32, 768, 256, 12, 64, 32, 2, 0.75
chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
eng = LTMEngine(N, H, DH, D, P, tau=TAU, sticky=True, n_layers=L, max_q=Q, device=dev, max_batch_chunks=42)
projs = [tuple(torch.from_numpy(a).to(dev) for a in synth.layer_projections(l, D, H * DH)) for l in range(L)]
q = torch.from_numpy(np.stack([synth.layer_query(l, Q, H * DH) for l in range(L)])).to(dev)
u = torch.from_numpy(synth.gibbs_uniforms(chunks, L)).to(dev)
k = torch.empty(chunks, T * P, D, device=dev)
gen = torch.Generator(device=dev).manual_seed(1234)
for i in range(0, chunks, 64):
    k[i:i + 64].normal_(generator=gen)
single = eng.consolidate(k, q, projs, u, new_doc=True).clone()
out = {"chunks": chunks, "what": "relative L2 distance to the single-stream run (synthetic N(0,1) tokens, headline shape)"}
for R in (2, 4, 8):
    parts = []
    for r in range(R):
        a, b = shard_range(chunks, R, r)
        parts.append(eng.consolidate(k[a:b], q, projs, u[a:b], new_doc=True).clone())
    sharded = torch.cat(parts)
    rel = lambda x, y: float((x - y).norm() / y.norm())
    per_chunk = ((sharded - single).flatten(1).norm(dim=1) / single.flatten(1).norm(dim=1)).cpu().numpy()
    first_of_blocks = [shard_range(chunks, R, r)[0] for r in range(1, R)]
    out[f"R={R}"] = {
        "mean_over_chunks_rel_l2": rel(sharded.mean(0), single.mean(0)),
        "per_chunk_rel_l2_median": float(np.median(per_chunk)),
        "per_chunk_rel_l2_max": float(per_chunk.max()),
        "chunks_identical_to_single_stream": int((per_chunk == 0).sum()),
        "rel_l2_of_first_chunk_after_each_boundary": [float(per_chunk[c]) for c in first_of_blocks],
    }
print(json.dumps(out, indent=1))
