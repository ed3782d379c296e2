"""Golden cases of the video Q-former path (encode_video counterpart).

Each case is a chain of chunks pushed through the reference's 2-layer video ``BertEncoder``
(short-term cross-attention + LTM + merge + query FFN) followed by ``llama_proj``.  Inputs and weights are
regenerated from ``infinite_video_amd.synth``; fixtures hold outputs only.  Before chunk c the torch CPU generator
is seeded with ``chunk_seed(case, c)``; LTM layer l of that chunk consumes uniforms [1024*l, 1024*l + 512).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List

import numpy as np
import torch

from infinite_video_amd import synth

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))


@dataclass
class QFCase:
    name: str
    N: int = 64
    tau: float = 0.75
    alpha: float = 0.9
    sticky: bool = True
    chunk_T: List[int] = field(default_factory=lambda: [8] * 4)
    seed_base: int = 9000
    proj_out: int = 512
    n_layers: int = 2
    n_query: int = 32
    hidden: int = 768
    P: int = 32
    xq_gain: float = 1.0           # multiplies the cross-attention query weights (peaked softmax)


QF_CASES = [
    QFCase("qf_small", chunk_T=[8, 8, 5, 8]),
    QFCase("qf_alpha1", alpha=1.0, chunk_T=[8, 8], seed_base=9100, proj_out=4096),          # LTM bypassed (Qformer.py:220-223)
    QFCase("qf_uniform", sticky=False, chunk_T=[8, 8, 8], seed_base=9200),
    QFCase("qf_peaked", chunk_T=[16, 16, 16], xq_gain=12.0, seed_base=9250),
    QFCase("qf_headline", N=256, chunk_T=[256, 256], seed_base=9300),
]


def chunk_seed(case: QFCase, chunk: int) -> int:
    return case.seed_base + chunk


def chunk_uniforms(case: QFCase, chunk: int) -> np.ndarray:
    """u[l, 512] float64 exactly as the reference's two LTM calls of this chunk draw them."""
    torch.manual_seed(chunk_seed(case, chunk))
    u = np.empty((case.n_layers, synth.NB_SAMPLES), np.float64)
    for l in range(case.n_layers):
        u[l] = torch.rand(synth.NB_SAMPLES, dtype=torch.float64).numpy()
        torch.rand(synth.NB_SAMPLES, dtype=torch.float64)
    return u


def qf_inputs(case: QFCase):
    frames = [synth.frame_tokens(c, T, case.P, case.hidden, seed=synth.SEED_K + case.seed_base)
              for c, T in enumerate(case.chunk_T)]
    weights = synth.video_qformer_weights(case.n_layers, case.hidden, 4 * case.hidden, case.hidden, case.n_query,
                                          case.proj_out, seed=synth.SEED_QF + case.seed_base)
    if case.xq_gain != 1.0:
        for l in range(case.n_layers):
            weights[f"bert.encoder.layer.{l}.crossattention.self.query.weight"] *= np.float32(case.xq_gain)
    return frames, weights


def qf_golden_path(case: QFCase) -> str:
    return os.path.join(GOLDEN_DIR, f"{case.name}.npz")


def load_qf_golden(case: QFCase):
    return np.load(qf_golden_path(case))
