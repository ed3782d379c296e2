#!/usr/bin/env python3
"""Per-chunk timing of the video Q-former path (encode_video counterpart) at the headline shape:
T=256 frames x 32 tokens x 768, 2 layers, N=256, alpha=0.9, sticky.  Secondary number, not bench.py's metric."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from infinite_video_amd import synth
from infinite_video_amd.video_qformer import InfVideoEncoder


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=64)
    ap.add_argument("--T", type=int, default=256)
    ap.add_argument("--alpha", type=float, default=0.9)
    ap.add_argument("--distinct", type=int, default=16, help="distinct chunk tensors cycled through")
    ap.add_argument("--batched", type=int, default=0, help="chunks of one layer-major whole-video call (0 = per-chunk mode)")
    ap.add_argument("--calls", type=int, default=5, help="timed whole-video calls (after one warm-up call)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    m = InfVideoEncoder(num_basis=256, tau=0.75, alpha=args.alpha, sticky=True)
    m.load_reference_state_dict(synth.video_qformer_weights())
    m = m.to(dev)
    if args.batched:
        Cn = args.batched
        frames = torch.randn(Cn, args.T * 32, 768, device=dev)
        u = torch.from_numpy(synth.gibbs_uniforms(Cn, 2)).to(dev)
        m.encode_frames_batch(frames, new_video=True, u=u)             # full-size warm-up: workspaces grow here
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.calls):
            t0 = time.perf_counter()
            m.encode_frames_batch(frames, new_video=True, u=u)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        dt = sorted(ts)[len(ts) // 2]
        flops = 2 * 2 * (2 * 384 * 768 * args.T * 32)
        print(json.dumps({"what": "encode_video counterpart, layer-major whole video (median of the timed calls)", "T": args.T,
                          "alpha": args.alpha, "chunks": Cn, "ms_per_chunk": 1e3 * dt / Cn, "chunks_per_s": Cn / dt,
                          "best_ms_per_chunk": 1e3 * min(ts) / Cn, "short_attention_tflops": flops * Cn / dt / 1e12}))
        return
    ks = [torch.randn(1, args.T * 32, 768, device=dev) for _ in range(args.distinct)]
    u = torch.from_numpy(synth.gibbs_uniforms(args.chunks + 4, 2)).to(dev)
    for c in range(4):
        m.encode_frames(ks[c % args.distinct], new_video=(c == 0), u=u[c])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for c in range(args.chunks):
        m.encode_frames(ks[c % args.distinct], new_video=False, u=u[4 + c])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    flops = 2 * 2 * (2 * 384 * 768 * args.T * 32)          # two [H*Q x d x T*P] contractions per layer
    print(json.dumps({"what": "encode_video counterpart, per chunk", "T": args.T, "alpha": args.alpha,
                      "ms_per_chunk": 1e3 * dt / args.chunks, "chunks_per_s": args.chunks / dt,
                      "short_attention_tflops": flops * args.chunks / dt / 1e12}))


if __name__ == "__main__":
    main()
