"""Generate tests/golden/*.npz by running the REAL reference operator on the CPU.

Build-container only: needs /root/reference (never present on the GPU box).  Run from the
repo root:   python tests/golden/make_goldens.py [case ...]

The reference modules are loaded by file path under a synthetic parent package so that
``InfVideoLLaMA/__init__.py`` (needs omegaconf) is bypassed; nothing of the reference is
copied into this repository -- only the numbers it produces.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import tempfile
import types

os.environ.setdefault("MPLBACKEND", "Agg")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from tests.golden.cases import CASES, DENSE_CASES, Case, call_seed, case_inputs, golden_path

REF = {
    "VL": "/root/reference/infty-Video-LLaMA/InfVideoLLaMA/models",
    "VC": "/root/reference/infty-VideoChat2/models/blip2",
}


def load_reference(variant: str):
    base = REF[variant]
    parent = "ref_" + variant.lower()
    if parent + ".long_term_attention_gibbs" in sys.modules:
        return sys.modules[parent + ".long_term_attention_gibbs"]
    pkg = types.ModuleType(parent)
    pkg.__path__ = [base]
    sys.modules[parent] = pkg
    for name in ("basis_functions", "long_term_attention_gibbs"):
        spec = importlib.util.spec_from_file_location(f"{parent}.{name}", os.path.join(base, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"{parent}.{name}"] = mod
        spec.loader.exec_module(mod)
    return sys.modules[parent + ".long_term_attention_gibbs"]


class _Recorder:
    """Stands in for ``torch.distributions`` inside the reference module and records what the
    sticky step feeds to / gets from ``Categorical`` (probabilities and drawn bins)."""

    def __init__(self):
        self.log = []
        rec = self

        class Categorical(torch.distributions.Categorical):
            def sample(self, sample_shape=torch.Size()):
                out = super().sample(sample_shape)
                rec.log.append((self.probs.detach().clone(), out.detach().clone()))
                return out

        self.Categorical = Categorical


def build_layer(mod, case: Case, proj):
    wk, bk, wv, bv = proj
    pk, pv = torch.nn.Linear(case.d, case.dm), torch.nn.Linear(case.d, case.dm)
    with torch.no_grad():
        pk.weight.copy_(torch.from_numpy(wk)); pk.bias.copy_(torch.from_numpy(bk))
        pv.weight.copy_(torch.from_numpy(wv)); pv.bias.copy_(torch.from_numpy(bv))
    # constructor kwargs exactly as the Q-former passes them (reference Qformer.py:135-158)
    return mod.LongTermAttention(
        head_size=case.dh, length=case.d, target_len=case.d, attn_func="softmax",
        attn_num_basis=case.N, continuous=True, attn_drop=0.1, infinite_memory=True, n_layers=2,
        n_heads=case.H, affines=True, mask=True, mask_type="cnn", kl_regularizer=False,
        proj_key=pk, proj_value=pv, sigma_0=None, mu_0=None, sticky_memories=case.sticky,
        sigmas=None, tau=case.tau, d_model=case.dm)


def sparse_rows(G: torch.Tensor):
    """Dense [rows, N] operator -> (col, val, nnz_per_row max)."""
    nz = (G != 0)
    col = torch.where(nz.any(1), nz.float().argmax(1), torch.full((G.size(0),), -1)).to(torch.int32)
    val = G.abs().max(1).values * torch.sign(G.sum(1))
    return col.numpy(), val.numpy().astype(np.float32), int(nz.sum(1).max())


def run_case(case: Case):
    mod = load_reference(case.variant)
    rec = _Recorder()
    mod.dist = rec
    ks, qs, ws = case_inputs(case)
    layers = [build_layer(mod, case, ws[l]) for l in range(case.n_layers)]
    out = {}
    with torch.no_grad():
        for c, T in enumerate(case.chunk_T):
            k = torch.from_numpy(ks[c]).unsqueeze(0)
            for l, m in enumerate(layers):
                q = torch.from_numpy(qs[l]).unsqueeze(0)
                torch.manual_seed(call_seed(case, c, l))
                n_log = len(rec.log)
                m.length = m.target_len = k.size(1)          # as the hook does, Qformer.py:218-219
                ctx = m(k, q, new_doc=(c in case.new_doc_at), layer_n=l)
                tag = f"c{c}_l{l}"
                out[tag + "_ctx"] = ctx[0].numpy().copy()
                B = m.B_past[0].numpy()
                if case.store_full_B:
                    out[tag + "_B"] = B.copy()
                else:
                    out[tag + "_Brows"] = B[::16].copy()
                    out[tag + "_Bsum"] = B.astype(np.float64).sum(1)
                # generator position after the call: the next float64 the reference would draw
                out[tag + "_next_u"] = torch.rand(1, dtype=torch.float64).numpy()
                new = rec.log[n_log:]
                if new:                                      # sticky draw happened: (p, b) then (ones, t)
                    out[tag + "_probs"] = new[0][0].reshape(-1).numpy().astype(np.float32)
                    out[tag + "_bins"] = new[0][1].reshape(-1).numpy().astype(np.int16)
                    assert int(new[1][1].abs().max()) == 0
                scores = (m.queries / (m.d_head ** 0.5)) @ m.keys.transpose(-1, -2)
                out[tag + "_scores"] = scores[0].numpy().astype(np.float32).copy() \
                    if case.store_full_B else scores[0, :, ::4, :].numpy().astype(np.float32).copy()
            # operators of this chunk length, from the first layer (identical across layers)
            if f"T{T}_first_col" not in out:
                m0 = layers[0]
                if case.dense:
                    # operators with two non-zeros in some rows: stored as the reference built them
                    out[f"T{T}_first_G"] = m0.Gs[T].cpu().numpy().astype(np.float32).copy()
                    out[f"T{T}_inf_G"] = m0.G_inf.cpu().numpy().astype(np.float32).copy()
                    out[f"T{T}_uniform_samples"] = m0.samples.cpu().numpy().astype(np.uint8).copy()
                    continue
                fc, fv, fn = sparse_rows(m0.Gs[T].cpu())
                ic, iv, inn = sparse_rows(m0.G_inf.cpu())
                assert fn <= 1 and inn <= 1, "reference operator is not one-nonzero-per-row"
                out[f"T{T}_first_col"], out[f"T{T}_first_val"] = fc, fv
                out[f"T{T}_inf_col"], out[f"T{T}_inf_val"] = ic, iv
                smp = m0.samples
                out[f"T{T}_uniform_idx"] = torch.where(
                    smp.sum(1) > 0, smp.argmax(1), torch.full((smp.size(0),), -1)).to(torch.int32).numpy()
    return out


def main(argv):
    names = set(argv[1:])
    work = tempfile.mkdtemp()       # the VL reference pickles ./alphas_uniform on every call
    os.chdir(work)
    for case in CASES + DENSE_CASES:
        if names and case.name not in names:
            continue
        out = run_case(case)
        np.savez_compressed(golden_path(case), **out)
        print(f"{case.name}: {len(out)} arrays, {os.path.getsize(golden_path(case)) / 1e6:.2f} MB")


if __name__ == "__main__":
    main(sys.argv)
