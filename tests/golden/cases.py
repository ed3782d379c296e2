"""Golden-vector cases shared by ``make_goldens.py`` (build container only: imports the
reference) and the tests (anywhere: read the committed ``.npz`` files).

A case is a chain of LTM forwards on one or more layers.  Inputs are regenerated from the
counter-based generator in ``infinite_video_amd.synth`` so fixtures only hold outputs.
Before the forward of (chunk c, layer l) the torch CPU generator is seeded with
``call_seed(case, c, l)``; the Gibbs uniforms of that call are therefore
``torch.rand(512, float64)`` right after the same seeding.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np
import torch

from infinite_video_amd import synth

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))


@dataclass
class Case:
    name: str
    variant: str = "VL"            # VL: P=32, d=768 ; VC: P=14*14, d=1024
    N: int = 64
    Q: int = 32
    tau: float = 0.75
    sticky: bool = True
    n_layers: int = 2
    chunk_T: List[int] = field(default_factory=lambda: [8] * 8)   # frames per chunk
    new_doc_at: Tuple[int, ...] = (0,)                               # chunks that start a document
    q_scale: float = 1.0
    seed_base: int = 1000
    store_full_B: bool = True      # else: every 16th row + per-row sums
    dense: bool = False            # num_basis whose fp32 boxes overlap: the reference's operators are stored dense

    @property
    def P(self): return 32 if self.variant == "VL" else 196
    @property
    def d(self): return 768 if self.variant == "VL" else 1024
    @property
    def pool_shape(self): return (32,) if self.variant == "VL" else (14, 14)
    dm: int = 768
    H: int = 12
    dh: int = 64


CASES = [
    Case("cfg1_sticky"),                                              # BASELINE config 1 (numerics gate)
    Case("cfg1_uniform", sticky=False, seed_base=2000),
    Case("ragged_reset", chunk_T=[8, 8, 5, 8, 8, 7], new_doc_at=(0, 3), seed_base=3000),
    Case("peaked", q_scale=8.0, chunk_T=[16] * 4, seed_base=4000),
    Case("tau09", tau=0.9, N=128, chunk_T=[12] * 3, seed_base=4500, n_layers=1),
    Case("headline", N=256, chunk_T=[256] * 3, seed_base=5000, store_full_B=False),
    Case("headline_ragged", N=256, chunk_T=[256, 255, 100], seed_base=5500, n_layers=1, store_full_B=False),
    Case("vc_shape", variant="VC", Q=96, chunk_T=[16] * 3, seed_base=6000, n_layers=2),
    # num_basis not a power of two: box edges are fp32 linspace values, membership is NOT floor(t*N) (BASIS.py:248-250).
    # (For many such values -- 48, 80, 96, 192, ... -- neighbouring fp32 boxes overlap at a sample position, the
    # reference's G is no longer one-non-zero-per-row and basis_maps.build_plan refuses them; 144 is one that works.)
    Case("n144", N=144, chunk_T=[12, 12, 9, 12], seed_base=7000, n_layers=1),
]


# num_basis whose fp32 boxes overlap (or leave a gap) exactly where the step evaluates psi: two non-zeros in some rows of
# the reference's G, histogram edges and resampling points that lie in two boxes (or in none).  Separate list: the
# sparse-form tests iterate over CASES.
DENSE_CASES = [
    Case("n96_dense", N=96, chunk_T=[16, 16, 12, 16], seed_base=8000, n_layers=2, dense=True),
    Case("n48_uniform_dense", N=48, sticky=False, chunk_T=[8, 8, 8], seed_base=8500, n_layers=1, dense=True),
    # any --num_basis, as the reference takes it (run_inference_inf_video_llama_nextqa.py:61): not a multiple of the kernels'
    # 16-wide tile (the device pads with inert basis functions) ...
    Case("n100_any", N=100, chunk_T=[8, 8, 6, 8], seed_base=9500, n_layers=2, dense=True),
    # ... and one whose 1000-point read-out grid has a point in two fp32 boxes (no count-weighted closed form: the general-psi
    # step with the rectangular psi as dense 0/1 rows)
    Case("n37_grid_overlap", N=37, chunk_T=[8, 8, 8], seed_base=9700, n_layers=1, dense=True),
]


# The reference's GAUSSIAN basis family as a whole LTM chain (make_gaussian_goldens.py points the module's builder hook at its
# own add_gaussian_basis_functions, long_term_attention_gibbs.py:167-174): every operator dense, psi(bins[b]) a dense row,
# the edge scores and the 1000-point read-out dense contractions.
GAUSS_SIGMAS = [0.03, 0.1]
GAUSS_CASES = [
    Case("gauss_chain", N=64, chunk_T=[16, 16, 12, 16], seed_base=9100, n_layers=2, dense=True),
    Case("gauss_uniform", N=64, sticky=False, chunk_T=[8, 8, 8], seed_base=9300, n_layers=1, dense=True),
]


def call_seed(case: Case, chunk: int, layer: int) -> int:
    return case.seed_base + 16 * chunk + layer


def call_uniforms(case: Case, chunk: int, layer: int) -> np.ndarray:
    torch.manual_seed(call_seed(case, chunk, layer))
    return torch.rand(synth.NB_SAMPLES, dtype=torch.float64).numpy()


def case_inputs(case: Case):
    """(k per chunk, q per layer, projections per layer) as numpy fp32."""
    ks = [synth.frame_tokens(c, T, case.P, case.d, seed=synth.SEED_K + case.seed_base)
          for c, T in enumerate(case.chunk_T)]
    qs = [synth.layer_query(l, case.Q, case.dm, seed=synth.SEED_Q + case.seed_base, scale=case.q_scale)
          for l in range(case.n_layers)]
    ws = [synth.layer_projections(l, case.d, case.dm, seed=synth.SEED_W + case.seed_base)
          for l in range(case.n_layers)]
    return ks, qs, ws


def golden_path(case: Case) -> str:
    return os.path.join(GOLDEN_DIR, f"{case.name}.npz")


def load_golden(case: Case):
    return np.load(golden_path(case))
