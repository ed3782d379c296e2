"""Generate tests/golden/vc_mistral.npz by running the REAL VideoChat2 Q-former encoder of the reference on the CPU.

Build-container only (needs /root/reference).  Run from the repo root:
    python tests/golden/make_vc_goldens.py

What runs is the reference's own ``BertEncoder`` (infty-VideoChat2/models/blip2/Qformer.py:537-640) configured the way
``Blip2Base.init_Qformer`` does (blip2.py:47-77: bert-base, ``cross_attention_freq = 2``, ``encoder_width`` = the
vision width) with its own ``LongTermAttention`` (blip2/long_term_attention_gibbs.py, 14x14x1024 pooling) inside every
cross-attention, then a ``Linear`` standing for ``mistral_proj`` applied to the query part
(videochat2_it_mistral.py:252), chunk by chunk as ``infer_egoschema_inf`` drives it (run_nextqa_mistral.py:141-152).
``BertModel`` itself cannot be constructed under the installed transformers and is not needed:
with all-ones masks its forward adds zero masks and calls the embeddings and this encoder.
Nothing of the reference is copied into the repository -- only the numbers it produces.
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

from tests.golden.vc_cases import VC_CASE, VCCase, chunk_seed, vc_golden_path, vc_inputs

BASE = "/root/reference/infty-VideoChat2/models/blip2"


def load_reference_vc_qformer():
    name = "refvc_models.blip2.Qformer"
    if name in sys.modules:
        return sys.modules[name]
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    for n in ("apply_chunking_to_forward", "prune_linear_layer"):
        if not hasattr(mu, n):
            setattr(mu, n, getattr(pu, n))
    if not hasattr(mu, "find_pruneable_heads_and_indices"):
        mu.find_pruneable_heads_and_indices = getattr(pu, "find_pruneable_heads_and_indices", lambda *a, **k: None)
    for parent, path in (("refvc_models", os.path.dirname(BASE)), ("refvc_models.blip2", BASE)):
        pkg = types.ModuleType(parent)
        pkg.__path__ = [path]
        sys.modules[parent] = pkg
    for n in ("basis_functions", "long_term_attention_gibbs", "Qformer"):
        full = f"refvc_models.blip2.{n}"
        spec = importlib.util.spec_from_file_location(full, os.path.join(BASE, n + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[full] = mod
        spec.loader.exec_module(mod)
    return sys.modules[name]


def build(case: VCCase, weights):
    qf = load_reference_vc_qformer()
    cfg = qf.BertConfig()                        # defaults == bert-base-uncased (no network for from_pretrained)
    cfg.num_hidden_layers = case.n_layers        # blip2.py:55-66
    cfg.encoder_width = case.enc_width
    cfg.add_cross_attention = True
    cfg.cross_attention_freq = case.cross_freq
    cfg.query_length = case.n_query
    cfg.sticky = case.sticky
    cfg.num_basis = case.N
    cfg.tau = case.tau
    cfg.alpha = case.alpha
    enc = qf.BertEncoder(cfg)
    proj = torch.nn.Linear(case.hidden, case.proj_out)
    sd = {k[len("bert.encoder."):]: torch.from_numpy(v) for k, v in weights.items() if k.startswith("bert.encoder.")}
    missing = enc.load_state_dict(sd, strict=False)
    # long_term_attention.proj_{key,value} alias the layer's own key/value Linear (Qformer.py:156-157)
    assert not missing.unexpected_keys and all(".long_term_attention.proj_" in k for k in missing.missing_keys), missing
    proj.load_state_dict({"weight": torch.from_numpy(weights["mistral_proj.weight"]),
                          "bias": torch.from_numpy(weights["mistral_proj.bias"])})
    return enc.eval(), proj.eval()


def run_case(case: VCCase):
    frames, h0, weights = vc_inputs(case)
    enc, proj = build(case, weights)
    out, taps = {}, {}

    def tap(name):
        def hook(mod, args, output):
            taps[name] = (output[0] if isinstance(output, tuple) else output).detach().clone()
        return hook

    cross_layers = [l for l in range(case.n_layers) if l % case.cross_freq == 0]
    for l in cross_layers:
        layer = enc.layer[l]
        layer.crossattention.self.register_forward_hook(tap(f"l{l}_xctx"))             # merged context (:302-303)
        layer.crossattention.self.query.register_forward_hook(tap(f"l{l}_xq"))          # mixed_query_layer (:209)
        layer.crossattention.self.long_term_attention.register_forward_hook(tap(f"l{l}_along"))
    video = torch.from_numpy(frames)                                                    # [F, P, enc_width]
    chunks = torch.chunk(video, case.num_samples, dim=0)                                # run_nextqa_mistral.py:141
    embs = []
    with torch.no_grad():
        for c, blk in enumerate(chunks):
            k = blk.reshape(1, -1, case.enc_width)                                      # [1, T*196, 1024]  (:196)
            torch.manual_seed(chunk_seed(case, c))
            taps.clear()
            hid = torch.from_numpy(h0).unsqueeze(0)
            n_tok = hid.size(1)
            res = enc(hid, None, attention_mask=torch.zeros(1, 1, 1, n_tok), head_mask=[None] * case.n_layers,
                      encoder_hidden_states=k, encoder_attention_mask=torch.zeros(1, 1, 1, k.size(1)), return_dict=True,
                      query_length=case.n_query, new_video=(c == 0))
            last = res.last_hidden_state
            emb = proj(last[:, :case.n_query, :])                                       # videochat2_it_mistral.py:252
            embs.append(emb)
            out[f"c{c}_mistral"] = emb[0].numpy().copy()
            out[f"c{c}_next_u"] = torch.rand(1, dtype=torch.float64).numpy()
            if c in (0, 1, len(chunks) - 1):
                out[f"c{c}_hidden"] = last[0].numpy().copy()
            if c in (1, len(chunks) - 1):
                for l in (cross_layers[0], cross_layers[-1]):
                    for nm in ("xq", "along", "xctx"):
                        out[f"c{c}_l{l}_{nm}"] = taps[f"l{l}_{nm}"][0].numpy().copy()
            for l in cross_layers:
                B = enc.layer[l].crossattention.self.long_term_attention.B_past
                out[f"c{c}_l{l}_Bsum"] = B[0].numpy().astype(np.float64).sum(1)
        out["mean_mistral"] = torch.mean(torch.stack(embs), dim=0, keepdim=True).squeeze(0)[0].numpy().copy()   # :152
    return out


def main():
    os.chdir(tempfile.mkdtemp())
    case = VC_CASE
    out = run_case(case)
    np.savez_compressed(vc_golden_path(case), **out)
    print(f"{case.name}: {len(out)} arrays, {os.path.getsize(vc_golden_path(case)) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
