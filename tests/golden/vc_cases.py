"""Golden case of the VideoChat2 binding (BASELINE configs[4]): the reference's 12-layer
Q-former encoder (cross-attention every 2nd layer -> 6 LongTermAttention instances, ``sigmas=1``, the hook fires on
every cross-attention: infty-VideoChat2/models/blip2/Qformer.py:215-222,302-303) over ``num_samples`` chunks of
``max_int`` frames of 14x14 UMT-L patches (width 1024), 96 query tokens + instruction text tokens, followed by
``mistral_proj`` on the query part (videochat2_it_mistral.py:181-253) and the eval loop's mean over chunks
(eval_code/run_nextqa_mistral.py:141-152).

Inputs and weights are regenerated from ``infinite_video_amd.synth``; the fixture holds outputs only.  The embedding
layer (word/position embeddings of the instruction, upstream BERT plumbing) is not part of the case: the encoder is
fed its output directly -- LayerNorm-scaled synthetic rows for the query tokens and the text tokens.
Before chunk c the torch CPU generator is seeded with ``chunk_seed(c)``; LTM instance j (cross layer 2j) of that chunk
consumes uniforms [1024*j, 1024*j + 512)."""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np
import torch

from infinite_video_amd import synth

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))


@dataclass
class VCCase:
    name: str = "vc_mistral"
    N: int = 64                 # num_basis
    tau: float = 0.75
    alpha: float = 0.75         # videochat2_it_mistral.py:43
    sticky: bool = True
    n_layers: int = 12
    cross_freq: int = 2
    n_query: int = 96           # 32 query + 64 extra query tokens (configs/config_mistral.json)
    n_text: int = 8             # instruction tokens that ride along in the self-attention
    hidden: int = 768
    enc_width: int = 1024
    P: int = 196
    max_int: int = 16           # frames per chunk
    num_samples: int = 8        # chunks
    proj_out: int = 64
    seed_base: int = 12000

    @property
    def n_ltm(self):
        return (self.n_layers + self.cross_freq - 1) // self.cross_freq


VC_CASE = VCCase()


def chunk_seed(case: VCCase, chunk: int) -> int:
    return case.seed_base + chunk


def chunk_uniforms(case: VCCase, chunk: int) -> np.ndarray:
    """u[j, 512] float64 exactly as the reference's six LTM calls of this chunk draw them."""
    torch.manual_seed(chunk_seed(case, chunk))
    u = np.empty((case.n_ltm, synth.NB_SAMPLES), np.float64)
    for j in range(case.n_ltm):
        u[j] = torch.rand(synth.NB_SAMPLES, dtype=torch.float64).numpy()
        torch.rand(synth.NB_SAMPLES, dtype=torch.float64)
    return u


def vc_inputs(case: VCCase):
    """(frame tokens of the whole video [F, P, enc_width], encoder input rows [n_query + n_text, hidden], weights)."""
    F = case.max_int * case.num_samples
    frames = np.stack([synth.frame_tokens(f, 1, case.P, case.enc_width, seed=synth.SEED_K + case.seed_base)
                       for f in range(F)]).reshape(F, case.P, case.enc_width)
    weights = synth.videochat2_qformer_weights(case.n_layers, case.hidden, 4 * case.hidden, case.enc_width, case.cross_freq,
                                               case.n_query, case.proj_out, seed=synth.SEED_VC + case.seed_base)
    text = synth.layer_query(7, case.n_text, case.hidden, seed=synth.SEED_Q + case.seed_base)
    h0 = np.concatenate([weights["query_tokens"][0] * np.float32(50.0), text], 0).astype(np.float32)   # O(1) rows, as after a LayerNorm
    return frames, h0, weights


def vc_golden_path(case: VCCase) -> str:
    return os.path.join(GOLDEN_DIR, f"{case.name}.npz")


def load_vc_golden(case: VCCase):
    return np.load(vc_golden_path(case))
