#!/usr/bin/env python3
"""Run the same consolidate call several times with the draw trace on; report where runs part (chunk, layer, what)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tests.test_timed_path_gpu import _engine, _video, L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
k, q, projs, u, ws, qs = _video(dev, n)
eng = _engine(dev, max_batch_chunks=42)
out = []
for r in range(runs):
    bins, probs = eng.set_trace(n)
    ctx = eng.consolidate(k, q, projs, u, new_doc=True)
    eng.sync()
    out.append((ctx.cpu().numpy().copy(), bins.cpu().numpy().copy(), probs.cpu().numpy().copy()))
    eng.set_trace(0)
ref = out[0]
for r in range(1, runs):
    c, b, p = out[r]
    dc = np.abs(c - ref[0]).reshape(n, -1).max(1)
    db = (b != ref[1]).reshape(n, -1).sum(1)
    dp = np.abs(p - ref[2]).reshape(n, -1).max(1)
    bad = np.nonzero((dc > 0) | (db > 0) | (dp > 0))[0]
    print(f"run {r}: differing chunks {bad[:12].tolist()} (of {len(bad)})")
    for cc in bad[:4]:
        print(f"   chunk {cc}: max|dctx| {dc[cc]:.3e}  bins differing {db[cc]}  max|dprobs| {dp[cc]:.3e}")
        for l in range(L):
            w = np.nonzero(p[cc, l] != ref[2][cc, l])[0]
            print(f"      layer {l}: probs differ at bins {w[:10].tolist()} (n={len(w)})  e.g. {p[cc, l][w[:3]]} vs {ref[2][cc, l][w[:3]]}")
print("nan in ctx:", bool(np.isnan(ref[0]).any()))
