"""Parity of the HIP path (through the C ABI) with the CPU oracle and the reference goldens.
Needs a real MI355X: run with ``-m gpu``."""
import numpy as np
import pytest
import torch

from oracle import ltm_oracle as O
from tests.golden.cases import CASES, call_uniforms, case_inputs, load_golden

pytestmark = pytest.mark.gpu

CTX_TOL = 1e-4      # north-star tolerance is 1e-3 fp32; the kernels are held to 10x tighter
B_TOL = 2e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _engine(case, dev, n_layers=None, **kw):
    from infinite_video_amd.engine import LTMEngine
    return LTMEngine(case.N, case.H, case.dh, case.d, case.P, tau=case.tau, sticky=case.sticky,
                     n_layers=n_layers or case.n_layers, max_q=case.Q, device=dev, **kw)


def _to(dev, *arrs):
    return [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrs]


def _oracles(case, ws):
    return [O.ClosedFormOracle(case.N, case.H, case.dh, case.tau, case.sticky, *ws[l], tokens_per_frame=case.P)
            for l in range(case.n_layers)]


def _golden_B_check(case, g, tag, B):
    if case.store_full_B:
        np.testing.assert_allclose(B, g[tag + "_B"], rtol=0, atol=B_TOL)
    else:
        np.testing.assert_allclose(B[::16], g[tag + "_Brows"], rtol=0, atol=B_TOL)


@pytest.mark.parametrize("P,d,T", [(32, 768, 8), (32, 768, 256), (196, 1024, 16), (32, 768, 5)])
def test_pool_matches_oracle(dev, P, d, T):
    from infinite_video_amd.engine import LTMEngine
    eng = LTMEngine(64, 12, 64, d, P, tau=.75, sticky=True, device=dev)
    k = np.random.default_rng(T).standard_normal((T * P, d), dtype=np.float32)
    out = eng.pool(torch.from_numpy(k).to(dev)).cpu().numpy()
    np.testing.assert_allclose(out, O.ClosedFormOracle.pool(k, P), rtol=0, atol=2e-6)


@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_chain_free_running_matches_oracle_and_reference(dev, case):
    """Per-chunk forward() of all layers; the GPU derives its own sticky probabilities."""
    g = load_golden(case)
    ks, qs, ws = case_inputs(case)
    eng = _engine(case, dev)
    projs = [tuple(_to(dev, *w)) for w in ws]
    q = torch.from_numpy(np.stack(qs)).to(dev)
    orc = _oracles(case, ws)
    for c in range(len(case.chunk_T)):
        new_doc = c in case.new_doc_at
        u = np.stack([call_uniforms(case, c, l) for l in range(case.n_layers)])
        ctx = eng.forward(torch.from_numpy(ks[c]).to(dev), q, projs, torch.from_numpy(u).to(dev), new_doc=new_doc)
        ctx = ctx.cpu().numpy()
        for l in range(case.n_layers):
            tag = f"c{c}_l{l}"
            ref = orc[l].step(ks[c], qs[l], new_doc=new_doc, u=u[l])
            if case.sticky and not new_doc:
                bins, idx, probs = eng.last_draw(l)
                np.testing.assert_allclose(probs, orc[l].last_probs, rtol=2e-5, atol=1e-9)
                assert (bins == orc[l].last_bins).all(), \
                    f"{tag}: {(bins != orc[l].last_bins).sum()} of 512 Gibbs draws differ (free-running)"
                np.testing.assert_array_equal(bins, g[tag + "_bins"])      # == the reference's own draw
                np.testing.assert_array_equal(idx, orc[l].last_idx)
            B, _ = eng.export_state(l)
            np.testing.assert_allclose(B.cpu().numpy(), orc[l].B_past, rtol=0, atol=B_TOL)
            _golden_B_check(case, g, tag, B.cpu().numpy())
            np.testing.assert_allclose(ctx[l], ref, rtol=0, atol=CTX_TOL)
            np.testing.assert_allclose(ctx[l], g[tag + "_ctx"], rtol=0, atol=CTX_TOL)
            sc = eng.last_scores(l, case.Q)
            np.testing.assert_allclose(sc, orc[l].S_prev, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("case", [c for c in CASES if c.sticky], ids=lambda c: c.name)
def test_gibbs_draw_bit_exact_when_teacher_forced(dev, case):
    """Given the oracle's probabilities, the drawn bins must equal torch.multinomial's exactly."""
    g = load_golden(case)
    ks, qs, ws = case_inputs(case)
    eng = _engine(case, dev)
    projs = [tuple(_to(dev, *w)) for w in ws]
    q = torch.from_numpy(np.stack(qs)).to(dev)
    orc = _oracles(case, ws)
    for c in range(len(case.chunk_T)):
        new_doc = c in case.new_doc_at
        u = np.stack([call_uniforms(case, c, l) for l in range(case.n_layers)])
        refs = [orc[l].step(ks[c], qs[l], new_doc=new_doc, u=u[l]) for l in range(case.n_layers)]
        if not new_doc:
            for l in range(case.n_layers):
                eng.set_probs(l, g[f"c{c}_l{l}_probs"])                    # the reference's own p
        ctx = eng.forward(torch.from_numpy(ks[c]).to(dev), q, projs, torch.from_numpy(u).to(dev), new_doc=new_doc)
        for l in range(case.n_layers):
            if not new_doc:
                bins, idx, probs = eng.last_draw(l)
                np.testing.assert_array_equal(probs, g[f"c{c}_l{l}_probs"])
                np.testing.assert_array_equal(bins, g[f"c{c}_l{l}_bins"])
            np.testing.assert_allclose(ctx[l].cpu().numpy(), refs[l], rtol=0, atol=CTX_TOL)


def test_draw_kernel_equals_torch_multinomial_on_skewed_probs(dev):
    """Adversarial probabilities, many uniforms: every draw identical to torch.multinomial (CPU)."""
    case = CASES[0]
    ks, qs, ws = case_inputs(case)
    eng = _engine(case, dev, n_layers=1)
    projs = [tuple(_to(dev, *ws[0]))]
    q = torch.from_numpy(qs[0][None]).to(dev)
    eng.forward(torch.from_numpy(ks[0]).to(dev), q, projs, None, new_doc=True)
    for trial in range(6):
        gen = torch.Generator().manual_seed(trial)
        p = O.categorical_probs(torch.rand(127, generator=gen).pow(1 + 2 * trial) + 1e-8)
        torch.manual_seed(900 + trial)
        expect = torch.multinomial(p, 512, True).numpy()
        torch.manual_seed(900 + trial)
        u = torch.rand(512, dtype=torch.float64)
        eng.set_probs(0, p.numpy())
        eng.forward(torch.from_numpy(ks[1]).to(dev), q, projs, u[None].to(dev), new_doc=False)
        bins, _, _ = eng.last_draw(0)
        np.testing.assert_array_equal(bins, expect)


@pytest.mark.parametrize("case", [CASES[0], CASES[1], CASES[5]], ids=lambda c: c.name)
def test_consolidate_equals_per_chunk_forward(dev, case):
    """The batched whole-video entry point must reproduce the per-chunk chain."""
    ks, qs, ws = case_inputs(case)
    projs = [tuple(_to(dev, *w)) for w in ws]
    q = torch.from_numpy(np.stack(qs)).to(dev)
    Cn = len(case.chunk_T)
    u = np.stack([np.stack([call_uniforms(case, c, l) for l in range(case.n_layers)]) for c in range(Cn)])
    ud = torch.from_numpy(u).to(dev)
    a = _engine(case, dev)
    per_chunk = torch.stack([a.forward(torch.from_numpy(ks[c]).to(dev), q, projs, ud[c], new_doc=(c == 0))
                             for c in range(Cn)])
    b = _engine(case, dev, max_batch_chunks=3)
    batched = b.consolidate(torch.from_numpy(np.stack(ks)).to(dev), q, projs, ud, new_doc=True)
    np.testing.assert_allclose(batched.cpu().numpy(), per_chunk.cpu().numpy(), rtol=0, atol=1e-5)
    for l in range(case.n_layers):
        np.testing.assert_allclose(b.export_state(l)[0].cpu().numpy(), a.export_state(l)[0].cpu().numpy(),
                                   rtol=0, atol=1e-6)
        np.testing.assert_array_equal(b.last_draw(l)[0], a.last_draw(l)[0])


def test_state_export_import_roundtrip(dev):
    case = CASES[0]
    ks, qs, ws = case_inputs(case)
    projs = [tuple(_to(dev, *w)) for w in ws]
    q = torch.from_numpy(np.stack(qs)).to(dev)
    us = [torch.from_numpy(np.stack([call_uniforms(case, c, l) for l in range(2)])).to(dev) for c in range(4)]
    a = _engine(case, dev)
    for c in range(3):
        a.forward(torch.from_numpy(ks[c]).to(dev), q, projs, us[c], new_doc=(c == 0))
    b = _engine(case, dev)
    for l in range(2):
        B, mass = a.export_state(l)
        b.import_state(l, B, mass, projs[l])
    ya = a.forward(torch.from_numpy(ks[3]).to(dev), q, projs, us[3], new_doc=False)
    yb = b.forward(torch.from_numpy(ks[3]).to(dev), q, projs, us[3], new_doc=False)
    for l in range(2):
        np.testing.assert_array_equal(a.last_draw(l)[0], b.last_draw(l)[0])
    np.testing.assert_allclose(ya.cpu().numpy(), yb.cpu().numpy(), rtol=0, atol=2e-5)


def test_ragged_query_length_and_errors(dev):
    from infinite_video_amd import _lib
    case = CASES[0]
    ks, qs, ws = case_inputs(case)
    eng = _engine(case, dev, n_layers=1)
    projs = [tuple(_to(dev, *ws[0]))]
    orc = _oracles(case, ws)[0]
    q20 = qs[0][:20]
    ctx = eng.forward(torch.from_numpy(ks[0]).to(dev), torch.from_numpy(q20[None]).to(dev), projs, None, new_doc=True)
    np.testing.assert_allclose(ctx[0].cpu().numpy(), orc.step(ks[0], q20, True), rtol=0, atol=CTX_TOL)
    u = call_uniforms(case, 1, 0)
    ctx = eng.forward(torch.from_numpy(ks[1]).to(dev), torch.from_numpy(q20[None]).to(dev), projs,
                      torch.from_numpy(u[None]).to(dev), new_doc=False)
    np.testing.assert_allclose(ctx[0].cpu().numpy(), orc.step(ks[1], q20, False, u=u), rtol=0, atol=CTX_TOL)
    assert (eng.last_draw(0)[0] == orc.last_bins).all()
    with pytest.raises(_lib.LTMError):        # sticky step on an existing memory without uniforms
        eng.forward(torch.from_numpy(ks[2]).to(dev), torch.from_numpy(q20[None]).to(dev), projs, None, new_doc=False)
    with pytest.raises(ValueError):
        eng.forward(torch.from_numpy(ks[2][:, :100].copy()).to(dev), torch.from_numpy(q20[None]).to(dev), projs)
