"""Operator-level goldens of the reference's GAUSSIAN basis family (build container only: imports /root/reference).

The active reference module only ever builds rectangular bases (``add_retangular_basis_functions`` is called at
long_term_attention_gibbs.py:101 and :268; ``add_gaussian_basis_functions`` at :167 has no caller).  Here the module's own
builder hook is pointed at its own Gaussian builder, so that ``get_basis`` / ``compute_G`` / ``value_function`` -- the REAL
reference code -- produce a fully dense ridge operator and the coefficients of a first chunk:

    G_first [T, N] = Gs[T]                         (compute_G with GaussianBasisFunctions.evaluate)
    B       [N, d] = value_function(kbar)          (B_past after forward(new_doc=True))

and whole sticky / uniform-resampling CHAINS with that family (update_inf :194-222, score :224-230, expected_value
:251-286 all run with Gaussian psi because GaussianBasisFunctions.batch_evaluate exists, basis_functions.py:161-164):
per chunk and layer ctx, full B, probabilities, drawn bins, scores -- recorded exactly as make_goldens.py records the
rectangular cases -- plus the dense operators of every chunk length.

Run from the repo root:  python tests/golden/make_gaussian_goldens.py   ->  tests/golden/gauss_operator.npz, gauss_chain.npz,
gauss_uniform.npz
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
from tests.golden.cases import GAUSS_CASES, GAUSS_SIGMAS, GOLDEN_DIR, Case, call_seed, case_inputs, golden_path
from tests.golden.make_goldens import _Recorder, build_layer, load_reference

CASE = Case("gauss_operator", N=64, chunk_T=[16], seed_base=9000, n_layers=1)
SIGMAS = GAUSS_SIGMAS


def run_chain(mod, case: Case):
    """make_goldens.run_case for a module whose builder hook builds Gaussian bases: same recorder, same tags."""
    rec = _Recorder()
    mod.dist = rec
    ks, qs, ws = case_inputs(case)
    layers = [build_layer(mod, case, ws[l]) for l in range(case.n_layers)]
    out = {"sigmas": np.asarray(SIGMAS, np.float32)}
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
                out[tag + "_B"] = m.B_past[0].numpy().copy()
                out[tag + "_next_u"] = torch.rand(1, dtype=torch.float64).numpy()
                new = rec.log[n_log:]
                if new:                                      # sticky draw happened: (p, b) then (ones, t)
                    out[tag + "_probs"] = new[0][0].reshape(-1).numpy().astype(np.float32)
                    out[tag + "_bins"] = new[0][1].reshape(-1).numpy().astype(np.int16)
                    assert int(new[1][1].abs().max()) == 0
                scores = (m.queries / (m.d_head ** 0.5)) @ m.keys.transpose(-1, -2)
                out[tag + "_scores"] = scores[0].numpy().astype(np.float32).copy()
            if f"T{T}_first_G" not in out:
                m0 = layers[0]
                out[f"T{T}_first_G"] = m0.Gs[T].cpu().numpy().astype(np.float32).copy()
                out[f"T{T}_inf_G"] = m0.G_inf.cpu().numpy().astype(np.float32).copy()
    return out


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
    for case in GAUSS_CASES:
        res = run_chain(mod, case)
        path = golden_path(case)
        np.savez_compressed(path, **res)
        fin = [k_ for k_ in res if k_.endswith("_ctx")]
        print(f"{case.name}: {len(res)} arrays, {os.path.getsize(path) / 1e6:.2f} MB, all finite:",
              all(bool(np.isfinite(res[k_]).all()) for k_ in res), "max |ctx|", max(float(np.abs(res[k_]).max()) for k_ in fin))


if __name__ == "__main__":
    main()
