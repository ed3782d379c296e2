"""Generate tests/golden/qf_*.npz by running the REAL reference video Q-former encoder on the CPU.

Build-container only (needs /root/reference).  Run from the repo root:
    python tests/golden/make_qformer_goldens.py [case ...]

What runs is the reference's own ``BertEmbeddings`` + ``BertEncoder`` (Qformer.py:55-113, 537-640) with its own
``LongTermAttention`` inside every cross-attention, configured the way ``init_video_Qformer`` does
(infinityqa.py:36-55, 202-209: 2 layers, cross_attention_freq=1, text FFN removed), then a ``Linear`` standing for
``llama_proj`` (infinityqa.py:342).  ``BertModel``/``BertLMHeadModel`` cannot be constructed under the installed
transformers, and are not needed: with an all-ones attention mask their forward adds
zero masks and calls exactly these two modules (Qformer.py:958-1013).
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

from tests.golden.qformer_cases import QF_CASES, QFCase, chunk_seed, qf_golden_path, qf_inputs

BASE = "/root/reference/infty-Video-LLaMA/InfVideoLLaMA/models"


def load_reference_qformer():
    if "InfVideoLLaMA.models.Qformer" in sys.modules:
        return sys.modules["InfVideoLLaMA.models.Qformer"]
    # names Qformer.py:41-46 imports from transformers.modeling_utils that moved in newer transformers
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    for name in ("apply_chunking_to_forward", "prune_linear_layer"):
        if not hasattr(mu, name):
            setattr(mu, name, getattr(pu, name))
    if not hasattr(mu, "find_pruneable_heads_and_indices"):
        mu.find_pruneable_heads_and_indices = getattr(pu, "find_pruneable_heads_and_indices", lambda *a, **k: None)
    for parent, path in (("InfVideoLLaMA", os.path.dirname(BASE)), ("InfVideoLLaMA.models", BASE)):
        pkg = types.ModuleType(parent)
        pkg.__path__ = [path]
        sys.modules[parent] = pkg
    for name in ("basis_functions", "long_term_attention_gibbs", "Qformer"):
        full = f"InfVideoLLaMA.models.{name}"
        spec = importlib.util.spec_from_file_location(full, os.path.join(BASE, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[full] = mod
        spec.loader.exec_module(mod)
    return sys.modules["InfVideoLLaMA.models.Qformer"]


def build(case: QFCase, weights):
    qf = load_reference_qformer()
    cfg = qf.BertConfig()                        # defaults == bert-base-uncased (no network for from_pretrained)
    cfg.num_hidden_layers = case.n_layers        # infinityqa.py:38-48
    cfg.encoder_width = case.hidden
    cfg.add_cross_attention = True
    cfg.cross_attention_freq = 1
    cfg.query_length = case.n_query
    cfg.sticky = case.sticky
    cfg.num_basis = case.N
    cfg.sigmas = [0.005, 0.01]
    cfg.tau = case.tau
    cfg.alpha = case.alpha
    emb = qf.BertEmbeddings(cfg)
    enc = qf.BertEncoder(cfg)
    emb.word_embeddings = None                   # infinityqa.py:203-208
    emb.position_embeddings = None
    for layer in enc.layer:
        layer.output = None
        layer.intermediate = None
    proj = torch.nn.Linear(case.hidden, case.proj_out)
    sd_emb = {k[len("bert.embeddings."):]: torch.from_numpy(v) for k, v in weights.items()
              if k.startswith("bert.embeddings.")}
    sd_enc = {k[len("bert.encoder."):]: torch.from_numpy(v) for k, v in weights.items()
              if k.startswith("bert.encoder.")}
    missing = emb.load_state_dict(sd_emb, strict=False)
    assert not missing.unexpected_keys and set(missing.missing_keys) <= {"position_ids"}, missing
    missing = enc.load_state_dict(sd_enc, strict=False)
    # long_term_attention.proj_{key,value} alias the layer's own key/value Linear (Qformer.py:156-157)
    assert not missing.unexpected_keys and all(".long_term_attention.proj_" in k for k in missing.missing_keys), missing
    proj.load_state_dict({"weight": torch.from_numpy(weights["llama_proj.weight"]),
                          "bias": torch.from_numpy(weights["llama_proj.bias"])})
    return emb.eval(), enc.eval(), proj.eval(), torch.from_numpy(weights["video_query_tokens"])


def run_case(case: QFCase):
    frames, weights = qf_inputs(case)
    emb, enc, proj, qtok = build(case, weights)
    out = {}
    taps = {}

    def tap(name):
        def hook(mod, args, output):
            taps[name] = (output[0] if isinstance(output, tuple) else output).detach().clone()
        return hook

    for l, layer in enumerate(enc.layer):
        layer.crossattention.self.register_forward_hook(tap(f"l{l}_xctx"))           # merged context (:303-304)
        layer.crossattention.self.query.register_forward_hook(tap(f"l{l}_xq"))        # mixed_query_layer (:211)
        layer.crossattention.self.long_term_attention.register_forward_hook(tap(f"l{l}_along"))

    with torch.no_grad():
        for c, T in enumerate(case.chunk_T):
            k = torch.from_numpy(frames[c]).unsqueeze(0)                              # [1, T*32, 768]
            flag = torch.zeros(1, k.size(1), 1)                                       # only tested for `is not None`
            torch.manual_seed(chunk_seed(case, c))
            taps.clear()
            h0 = emb(query_embeds=qtok)                                               # Qformer.py:942-947
            zeros_self = torch.zeros(1, 1, 1, case.n_query)                           # inverted all-ones masks
            zeros_enc = torch.zeros(1, 1, 1, k.size(1))
            res = enc(h0, flag, attention_mask=zeros_self, head_mask=[None] * case.n_layers,
                      encoder_hidden_states=k, encoder_attention_mask=zeros_enc, return_dict=True,
                      query_length=case.n_query, new_video=(c == 0))
            hid = res.last_hidden_state
            out[f"c{c}_hidden"] = hid[0].numpy().copy()
            out[f"c{c}_llama"] = proj(hid)[0].numpy().copy()
            out[f"c{c}_next_u"] = torch.rand(1, dtype=torch.float64).numpy()
            for name, v in taps.items():
                out[f"c{c}_{name}"] = v[0].numpy().copy()
            for l, layer in enumerate(enc.layer):
                B = layer.crossattention.self.long_term_attention.B_past
                if B is not None:
                    out[f"c{c}_l{l}_Bsum"] = B[0].numpy().astype(np.float64).sum(1)
        out["h0"] = h0[0].numpy().copy()
    return out


def main(argv):
    names = set(argv[1:])
    os.chdir(tempfile.mkdtemp())       # the reference LTM pickles ./alphas_uniform on every call
    for case in QF_CASES:
        if names and case.name not in names:
            continue
        out = run_case(case)
        np.savez_compressed(qf_golden_path(case), **out)
        print(f"{case.name}: {len(out)} arrays, {os.path.getsize(qf_golden_path(case)) / 1e6:.2f} MB")


if __name__ == "__main__":
    main(sys.argv)
