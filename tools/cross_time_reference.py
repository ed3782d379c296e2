#!/usr/bin/env python3
"""Build-container only: time the REAL reference operator (imported from /root/reference) against the CPU port
bench.py times on the GPU box (oracle.DenseOracle with the density side effect), same inputs, same thread counts.

    python tools/cross_time_reference.py [--threads 1 8] [--calls 6]

Prints one JSON object; the numbers quoted in BASELINE.md / DESIGN.md come from here.  The reference never
travels to the GPU box, so this is how the "port" row of cpu_baseline is tied to the reference's own speed."""
import argparse
import json
import os
import statistics
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MPLBACKEND", "Agg")
import torch

from infinite_video_amd import synth
from oracle.ltm_oracle import DenseOracle
from tests.golden.make_goldens import load_reference

T, P, D, N, H, DH, Q, TAU = 256, 32, 768, 256, 12, 64, 32, 0.75
DM = H * DH


def layers():
    wk, bk, wv, bv = synth.layer_projections(0, D, DM)
    pk, pv = torch.nn.Linear(D, DM), torch.nn.Linear(D, DM)
    with torch.no_grad():
        pk.weight.copy_(torch.from_numpy(wk)); pk.bias.copy_(torch.from_numpy(bk))
        pv.weight.copy_(torch.from_numpy(wv)); pv.bias.copy_(torch.from_numpy(bv))
    mod = load_reference("VL")
    ref = mod.LongTermAttention(
        head_size=DH, length=D, target_len=D, attn_func="softmax", attn_num_basis=N, continuous=True, attn_drop=0.1,
        infinite_memory=True, n_layers=2, n_heads=H, affines=True, mask=True, mask_type="cnn", kl_regularizer=False,
        proj_key=pk, proj_value=pv, sigma_0=None, mu_0=None, sticky_memories=True, sigmas=None, tau=TAU, d_model=DM)
    port = DenseOracle(N, H, DH, TAU, True, pk, pv, density_side_effect=True)
    return ref, port


def time_calls(fn, calls):
    ts = []
    for c in range(calls + 1):
        k = torch.from_numpy(synth.frame_tokens(c, T, P, D)).unsqueeze(0)
        t0 = time.perf_counter()
        fn(k, c == 0)
        if c > 0:                                  # call 0 is the new-document warm-up
            ts.append(time.perf_counter() - t0)
    return ts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, nargs="+", default=[1, 8])
    ap.add_argument("--calls", type=int, default=6)
    args = ap.parse_args()
    ref, port = layers()
    q = torch.from_numpy(synth.layer_query(0, Q, DM)).unsqueeze(0)
    out = {"shape": "T=256 P=32 d=768 N=256 Q=32 H=12, steady-state sticky calls of ONE layer", "torch": torch.__version__,
           "cpus": os.cpu_count(), "results": {}}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp, torch.no_grad():
        os.chdir(tmp)                              # the reference pickles ./alphas_uniform on every call
        try:
            for n in args.threads:
                torch.set_num_threads(n)
                torch.manual_seed(42)
                tr = time_calls(lambda k, nd: ref(k, q, new_doc=nd, layer_n=0), args.calls)
                torch.manual_seed(42)
                tp = time_calls(lambda k, nd: port.forward(k, q, new_doc=nd), args.calls)
                out["results"][str(n)] = {
                    "reference_ms_per_call": round(1e3 * statistics.median(tr), 1),
                    "port_ms_per_call": round(1e3 * statistics.median(tp), 1),
                    "port_over_reference": round(statistics.median(tp) / statistics.median(tr), 3),
                    "reference_chunks_per_s_2_layers": round(0.5 / statistics.median(tr), 3)}
        finally:
            os.chdir(cwd)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
