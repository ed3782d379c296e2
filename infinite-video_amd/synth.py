"""Synthetic inputs for the LTM consolidation path (SURVEY.md section 8d recipe).

Counter-based (numpy Philox keyed by ``(seed, index)``), so any rank can generate exactly
its own chunks and golden fixtures only need to store outputs.  Values:

* frame tokens   k_c   ~ N(0,1)      keyed (SEED_K, chunk c)        [T*P, d]
* layer queries  q_l   ~ N(0,1)      keyed (SEED_Q, layer l)        [Q, dm]
* projections    W_k, W_v ~ N(0, 0.02) (BERT initializer_range, reference Qformer.py:727),
                 biases ~ N(0, 0.02) keyed (SEED_W, layer l)  (the SURVEY recipe uses zero
                 biases; non-zero ones exercise the bias path and are what the goldens use)
* Gibbs uniforms u_{c,l}[512] float64 from ``torch.Generator().manual_seed(42)`` on the
  host, chunk-major / layer-minor, each followed by 512 discarded draws (the reference's
  degenerate in-bin draw, long_term_attention_gibbs.py:206).
"""
from __future__ import annotations

import numpy as np
import torch

SEED_K, SEED_Q, SEED_W, SEED_U = 1234, 4321, 99, 42
NB_SAMPLES = 512


def _normal(seed: int, index: int, shape, scale: float = 1.0) -> np.ndarray:
    gen = np.random.Generator(np.random.Philox(key=[seed, index]))
    out = gen.standard_normal(shape, dtype=np.float32)
    if scale != 1.0:
        out *= np.float32(scale)
    return out


def frame_tokens(chunk: int, T: int, P: int, d: int, seed: int = SEED_K) -> np.ndarray:
    """Tokens of one chunk, [T*P, d] fp32."""
    return _normal(seed, chunk, (T * P, d))


def layer_query(layer: int, Q: int, dm: int, seed: int = SEED_Q, scale: float = 1.0) -> np.ndarray:
    return _normal(seed, layer, (Q, dm), scale)


def layer_projections(layer: int, d: int, dm: int, seed: int = SEED_W, bias: bool = True):
    """(W_k [dm,d], b_k [dm], W_v [dm,d], b_v [dm]) of one cross-attention layer."""
    wk = _normal(seed, 4 * layer + 0, (dm, d), 0.02)
    wv = _normal(seed, 4 * layer + 1, (dm, d), 0.02)
    if bias:
        bk = _normal(seed, 4 * layer + 2, (dm,), 0.02)
        bv = _normal(seed, 4 * layer + 3, (dm,), 0.02)
    else:
        bk = np.zeros(dm, np.float32)
        bv = np.zeros(dm, np.float32)
    return wk, bk, wv, bv


def gibbs_uniforms(n_chunks: int, n_layers: int, seed: int = SEED_U, nb_samples: int = NB_SAMPLES) -> np.ndarray:
    """u[c, l, :] float64, in the order a single-process run of the reference would consume
    torch's CPU generator after ``torch.manual_seed(seed)``.  The first chunk of a document
    draws nothing in the reference; entries for it are still generated (and ignored) so that
    u is indexable by global chunk id regardless of how chunks are sharded."""
    gen = torch.Generator().manual_seed(seed)
    u = torch.empty(n_chunks, n_layers, nb_samples, dtype=torch.float64)
    for c in range(n_chunks):
        for l in range(n_layers):
            u[c, l] = torch.rand(nb_samples, dtype=torch.float64, generator=gen)
            torch.rand(nb_samples, dtype=torch.float64, generator=gen)
    return u.numpy()


SEED_QF = 777


def video_qformer_weights(n_layers: int = 2, hidden: int = 768, inter: int = 3072, enc_width: int = 768,
                          n_query: int = 32, proj_out: int = 4096, seed: int = SEED_QF) -> dict:
    """Random-init weights of the video Q-former + ``llama_proj`` under the reference's own
    state-dict names (infinityqa.py:195-209 builds it; Qformer.py:115-470 names the parameters):
    ``bert.embeddings.LayerNorm.*``, ``bert.encoder.layer.{l}.attention.{self.query|self.key|self.value|output.dense|
    output.LayerNorm}.*``, ``...crossattention...``, ``...intermediate_query.dense.*``, ``...output_query.{dense|LayerNorm}.*``,
    plus ``video_query_tokens`` [1, n_query, hidden] and ``llama_proj.{weight,bias}``.
    Linear weights ~ N(0, 0.02) (BERT initializer_range), biases ~ N(0, 0.02), LayerNorm gamma ~ 1 + N(0, 0.1),
    beta ~ N(0, 0.1) -- non-trivial on purpose so every term of the path is exercised."""
    out, idx = {}, [0]

    def nrm(shape, scale, shift=0.0):
        a = _normal(seed, idx[0], shape, scale)
        idx[0] += 1
        return a + np.float32(shift) if shift else a

    def linear(name, n_out, n_in):
        out[name + ".weight"] = nrm((n_out, n_in), 0.02)
        out[name + ".bias"] = nrm((n_out,), 0.02)

    def lnorm(name, n):
        out[name + ".weight"] = nrm((n,), 0.1, 1.0)
        out[name + ".bias"] = nrm((n,), 0.1)

    out["video_query_tokens"] = nrm((1, n_query, hidden), 0.02)
    lnorm("bert.embeddings.LayerNorm", hidden)
    for l in range(n_layers):
        p = f"bert.encoder.layer.{l}."
        for att, kv_in in (("attention", hidden), ("crossattention", enc_width)):
            linear(p + att + ".self.query", hidden, hidden)
            linear(p + att + ".self.key", hidden, kv_in)
            linear(p + att + ".self.value", hidden, kv_in)
            linear(p + att + ".output.dense", hidden, hidden)
            lnorm(p + att + ".output.LayerNorm", hidden)
        linear(p + "intermediate_query.dense", inter, hidden)
        linear(p + "output_query.dense", hidden, inter)
        lnorm(p + "output_query.LayerNorm", hidden)
    linear("llama_proj", proj_out, hidden)
    return out


SEED_VC = 888


def videochat2_qformer_weights(n_layers: int = 12, hidden: int = 768, inter: int = 3072, enc_width: int = 1024,
                               cross_freq: int = 2, n_query: int = 96, proj_out: int = 4096, seed: int = SEED_VC) -> dict:
    """Random-init weights of the VideoChat2 Q-former encoder + ``mistral_proj`` under the reference's state-dict names
    (infty-VideoChat2/models/blip2/Qformer.py:419-441: every layer has self-attention, the query FFN
    ``intermediate_query``/``output_query`` and the text FFN ``intermediate``/``output``; layers with
    ``l % cross_freq == 0`` also a cross-attention whose key/value read the ``enc_width``-wide frame tokens), plus
    ``query_tokens`` [1, n_query, hidden] (query + extra query tokens, videochat2_it_mistral.py:199-203).
    Same value conventions as :func:`video_qformer_weights`."""
    out, idx = {}, [0]

    def nrm(shape, scale, shift=0.0):
        a = _normal(seed, idx[0], shape, scale)
        idx[0] += 1
        return a + np.float32(shift) if shift else a

    def linear(name, n_out, n_in):
        out[name + ".weight"] = nrm((n_out, n_in), 0.02)
        out[name + ".bias"] = nrm((n_out,), 0.02)

    def lnorm(name, n):
        out[name + ".weight"] = nrm((n,), 0.1, 1.0)
        out[name + ".bias"] = nrm((n,), 0.1)

    out["query_tokens"] = nrm((1, n_query, hidden), 0.02)
    for l in range(n_layers):
        p = f"bert.encoder.layer.{l}."
        atts = [("attention", hidden)] + ([("crossattention", enc_width)] if l % cross_freq == 0 else [])
        for att, kv_in in atts:
            linear(p + att + ".self.query", hidden, hidden)
            linear(p + att + ".self.key", hidden, kv_in)
            linear(p + att + ".self.value", hidden, kv_in)
            linear(p + att + ".output.dense", hidden, hidden)
            lnorm(p + att + ".output.LayerNorm", hidden)
        for ffn in ("intermediate", "intermediate_query"):
            linear(p + ffn + ".dense", inter, hidden)
        for ffn in ("output", "output_query"):
            linear(p + ffn + ".dense", hidden, inter)
            lnorm(p + ffn + ".LayerNorm", hidden)
    linear("mistral_proj", proj_out, hidden)
    return out
