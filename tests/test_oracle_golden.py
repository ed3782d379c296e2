"""Pins oracle/ltm_oracle.py against golden vectors captured from the real reference
(tests/golden/make_goldens.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import ltm_oracle as O
from tests.golden.cases import (CASES, DENSE_CASES, GAUSS_CASES, GAUSS_SIGMAS, call_seed, call_uniforms, case_inputs,
                                load_golden)

FAST = [c for c in CASES if max(c.chunk_T) <= 16]
BIG = [c for c in CASES if max(c.chunk_T) > 16]


def _check_B(case, g, tag, B, atol):
    if case.store_full_B:
        np.testing.assert_allclose(B, g[tag + "_B"], rtol=0, atol=atol)
    else:
        np.testing.assert_allclose(B[::16], g[tag + "_Brows"], rtol=0, atol=atol)
        np.testing.assert_allclose(B.astype(np.float64).sum(1), g[tag + "_Bsum"], rtol=0, atol=atol * 100)


@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_maps_match_reference_operators(case):
    g = load_golden(case)
    for T in sorted(set(case.chunk_T)):
        mp = O.build_maps(T, case.N, case.tau)
        np.testing.assert_array_equal(mp.first_col, g[f"T{T}_first_col"])
        np.testing.assert_array_equal(mp.inf_col, g[f"T{T}_inf_col"])
        np.testing.assert_array_equal(mp.first_val, g[f"T{T}_first_val"])   # bit-exact 1/(c+ridge)
        np.testing.assert_array_equal(mp.inf_val, g[f"T{T}_inf_val"])
        np.testing.assert_array_equal(mp.uniform_idx, g[f"T{T}_uniform_idx"])
        assert mp.inf_col[-1] == -1          # t = 1.0 falls outside the last half-open box
        assert abs(float(mp.w.sum()) + mp.w_out - 1.0) < 1e-6


@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_closed_form_chain_matches_reference(case):
    g = load_golden(case)
    ks, qs, ws = case_inputs(case)
    layers = [O.ClosedFormOracle(case.N, case.H, case.dh, case.tau, case.sticky, *ws[l],
                                 tokens_per_frame=case.P) for l in range(case.n_layers)]
    for c in range(len(case.chunk_T)):
        for l, m in enumerate(layers):
            tag = f"c{c}_l{l}"
            u = call_uniforms(case, c, l)
            ctx = m.step(ks[c], qs[l], new_doc=(c in case.new_doc_at), u=u)
            if tag + "_bins" in g:
                np.testing.assert_array_equal(m.last_bins, g[tag + "_bins"].astype(np.int64))
                np.testing.assert_allclose(m.last_probs, g[tag + "_probs"], rtol=2e-6, atol=1e-9)
            else:
                assert (c in case.new_doc_at) or not case.sticky
            _check_B(case, g, tag, m.B_past, 2e-6)
            np.testing.assert_allclose(ctx, g[tag + "_ctx"], rtol=0, atol=2e-6)
            sc = m.S_prev if case.store_full_B else m.S_prev[:, ::4, :]
            np.testing.assert_allclose(sc, g[tag + "_scores"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("case", FAST + BIG[:1] + DENSE_CASES + GAUSS_CASES, ids=lambda c: c.name)
def test_dense_port_chain_matches_reference(case):
    """The reference-shaped port consumes torch's global generator like the reference.  DENSE_CASES: num_basis whose fp32
    boxes overlap (two non-zeros in rows of G, histogram edges in two boxes) -- the dense port is their oracle.
    GAUSS_CASES: the reference's Gaussian basis family as whole chains (tests/golden/make_gaussian_goldens.py)."""
    g = load_golden(case)
    gauss = case in GAUSS_CASES
    ks, qs, ws = case_inputs(case)
    layers = []
    for l in range(case.n_layers):
        wk, bk, wv, bv = ws[l]
        pk, pv = torch.nn.Linear(case.d, case.dm), torch.nn.Linear(case.d, case.dm)
        with torch.no_grad():
            pk.weight.copy_(torch.from_numpy(wk)); pk.bias.copy_(torch.from_numpy(bk))
            pv.weight.copy_(torch.from_numpy(wv)); pv.bias.copy_(torch.from_numpy(bv))
        layers.append(O.DenseOracle(case.N, case.H, case.dh, case.tau, case.sticky, pk, pv,
                                    pool_shape=case.pool_shape, gaussian_sigmas=GAUSS_SIGMAS if gauss else None))
    n_chunks = len(case.chunk_T) if (case in FAST or case.dense) else 2
    with torch.no_grad():
        for c in range(n_chunks):
            k = torch.from_numpy(ks[c]).unsqueeze(0)
            for l, m in enumerate(layers):
                tag = f"c{c}_l{l}"
                torch.manual_seed(call_seed(case, c, l))
                ctx = m.forward(k, torch.from_numpy(qs[l]).unsqueeze(0), new_doc=(c in case.new_doc_at))
                # same generator position afterwards as the reference
                np.testing.assert_array_equal(torch.rand(1, dtype=torch.float64).numpy(), g[tag + "_next_u"])
                if tag + "_bins" in g:
                    np.testing.assert_array_equal(m.last_bins.numpy(), g[tag + "_bins"].astype(np.int64))
                    np.testing.assert_array_equal(m.last_probs.reshape(-1).numpy(), g[tag + "_probs"])
                _check_B(case, g, tag, m.B_past[0].numpy(), 1e-7)
                np.testing.assert_allclose(ctx[0].numpy(), g[tag + "_ctx"], rtol=0, atol=1e-6)


def test_inverse_cdf_draw_is_torch_multinomial():
    """fp32 sequential cdf + lower-bound search == torch.multinomial on CPU, incl. skewed p."""
    for trial in range(4):
        gen = torch.Generator().manual_seed(trial)
        p = torch.rand(127, generator=gen).pow(1 + 3 * trial) + 1e-7
        p = O.categorical_probs(p)
        torch.manual_seed(77 + trial)
        b = torch.multinomial(p, 200000, True).numpy()
        torch.manual_seed(77 + trial)
        u = torch.rand(200000, dtype=torch.float64).numpy()
        np.testing.assert_array_equal(O.inverse_cdf_draw(p.numpy(), u), b)


def test_draw_uniforms_consumes_two_blocks():
    torch.manual_seed(5)
    u = O.draw_uniforms()
    nxt = torch.rand(1, dtype=torch.float64)
    torch.manual_seed(5)
    both = torch.rand(1024, dtype=torch.float64)
    np.testing.assert_array_equal(u, both[:512].numpy())
    torch.manual_seed(5)
    torch.rand(512, dtype=torch.float64); torch.rand(512, dtype=torch.float64)
    assert float(torch.rand(1, dtype=torch.float64)) == float(nxt)


def test_sticky_only_even_boxes_for_n256():
    _, edge_box, bin_box = O.sticky_bin_rows(256)
    assert edge_box[0] == -1 and edge_box[-1] == -1
    assert (bin_box[:127] == 2 * np.arange(127)).all()
