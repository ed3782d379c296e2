#!/usr/bin/env python3
"""Repeated create / consolidate / encode / destroy: device memory must return to where it started."""
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from infinite_video_amd import synth
from infinite_video_amd.engine import LTMEngine
from infinite_video_amd.video_qformer import InfVideoEncoder


def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2 ** 20


def main():
    dev = torch.device("cuda:0")
    N, H, dh, d, P, T, Q, L = 256, 12, 64, 768, 32, 256, 32, 2
    projs = [tuple(torch.from_numpy(a).to(dev) for a in synth.layer_projections(l, d, H * dh)) for l in range(L)]
    q = torch.from_numpy(np.stack([synth.layer_query(l, Q, H * dh) for l in range(L)])).to(dev)
    k = torch.randn(64, T * P, d, device=dev)
    u = torch.from_numpy(synth.gibbs_uniforms(64, L)).to(dev)
    w = synth.video_qformer_weights()
    base = None
    for it in range(6):
        eng = LTMEngine(N, H, dh, d, P, tau=0.75, sticky=True, n_layers=L, max_q=Q, device=dev, max_batch_chunks=42)
        for _ in range(5):
            eng.consolidate(k, q, projs, u, new_doc=True)
            eng.forward(k[0], q, projs, u[0], new_doc=False)
        m = InfVideoEncoder(num_basis=N, llama_hidden=512).to(dev)
        m.load_reference_state_dict({**w, "llama_proj.weight": w["llama_proj.weight"][:512], "llama_proj.bias": w["llama_proj.bias"][:512]})
        m.encode_frames_batch(k[:32], new_video=True, u=u[:32])
        m.encode_frames(k[:1], new_video=False, u=u[33])
        del eng, m
        gc.collect()
        torch.cuda.empty_cache()
        f = free_mb()
        if base is None:
            base = f
        print(f"iteration {it}: free {f:.0f} MiB (delta vs first {f - base:+.0f})", flush=True)
    assert abs(f - base) < 64, "device memory is not returned"


if __name__ == "__main__":
    main()
