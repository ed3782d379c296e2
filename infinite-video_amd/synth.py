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
