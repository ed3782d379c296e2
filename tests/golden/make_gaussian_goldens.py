"""Operator-level goldens of the reference's GAUSSIAN basis family (build container only: imports /root/reference).

The active reference module only ever builds rectangular bases (``add_retangular_basis_functions`` is called at
long_term_attention_gibbs.py:101 and :268; ``add_gaussian_basis_functions`` at :167 has no caller).  Here the module's own
builder hook is pointed at its own Gaussian builder, so that ``get_basis`` / ``compute_G`` / ``value_function`` -- the REAL
reference code -- produce a fully dense ridge operator and the coefficients of a first chunk:

    G_first [T, N] = Gs[T]                         (compute_G with GaussianBasisFunctions.evaluate)
    B       [N, d] = value_function(kbar)          (B_past after forward(new_doc=True))

Run from the repo root:  python tests/golden/make_gaussian_goldens.py   ->  tests/golden/gauss_operator.npz
"""
from __future__ import annotations

import os
import sys
import tempfile

os.environ.setdefault("MPLBACKEND", "Agg")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from infinite_video_amd import synth
from tests.golden.cases import GOLDEN_DIR, Case
from tests.golden.make_goldens import build_layer, load_reference

CASE = Case("gauss_operator", N=64, chunk_T=[16], seed_base=9000, n_layers=1)
SIGMAS = [0.03, 0.1]


def main():
    mod = load_reference("VL")
    cls = mod.LongTermAttention
    # the module's builder hook -> its own Gaussian builder (same signature but for `sigmas`)
    cls.add_retangular_basis_functions = lambda self, psi, nb_basis, device: cls.add_gaussian_basis_functions(
        self, psi, nb_basis, SIGMAS, device)
    os.chdir(tempfile.mkdtemp())
    T, P, d = CASE.chunk_T[0], CASE.P, CASE.d
    k = synth.frame_tokens(0, T, P, d, seed=synth.SEED_K + CASE.seed_base)
    q = synth.layer_query(0, CASE.Q, CASE.dm, seed=synth.SEED_Q + CASE.seed_base)
    w = synth.layer_projections(0, d, CASE.dm, seed=synth.SEED_W + CASE.seed_base)
    m = build_layer(mod, CASE, w)
    with torch.no_grad():
        m.length = m.target_len = T * P
        m(torch.from_numpy(k).unsqueeze(0), torch.from_numpy(q).unsqueeze(0), new_doc=True, layer_n=0)
    out = {"sigmas": np.asarray(SIGMAS, np.float32), "T": np.int32(T), "N": np.int32(CASE.N),
           "G_first": m.Gs[T].cpu().numpy().astype(np.float32), "B": m.B_past[0].numpy().astype(np.float32)}
    path = os.path.join(GOLDEN_DIR, "gauss_operator.npz")
    np.savez_compressed(path, **out)
    print("gauss_operator:", {k_: v.shape for k_, v in out.items()}, f"{os.path.getsize(path) / 1e6:.2f} MB",
          "max |G|", float(np.abs(out["G_first"]).max()), "max |B|", float(np.abs(out["B"]).max()))


if __name__ == "__main__":
    main()
