"""The drop-in nn.Module / Q-former hook against the reference goldens.  Needs an MI355X."""
import types

import numpy as np
import pytest
import torch

from tests.golden.cases import CASES, call_seed, case_inputs, load_golden

pytestmark = pytest.mark.gpu
CTX_TOL = 1e-4


def _linear(w, b, dev):
    lin = torch.nn.Linear(w.shape[1], w.shape[0])
    with torch.no_grad():
        lin.weight.copy_(torch.from_numpy(w)); lin.bias.copy_(torch.from_numpy(b))
    return lin.to(dev)


def _module(case, ws, l, dev):
    from infinite_video_amd.long_term_attention_gibbs import LongTermAttention, LongTermAttentionVC
    cls = LongTermAttention if case.variant == "VL" else LongTermAttentionVC
    wk, bk, wv, bv = ws[l]
    # constructor kwargs exactly as the reference Q-former passes them (Qformer.py:135-158)
    return cls(head_size=case.dh, length=case.d, target_len=case.d, attn_func="softmax",
               attn_num_basis=case.N, continuous=True, attn_drop=0.1, infinite_memory=True, n_layers=2,
               n_heads=case.H, affines=True, mask=True, mask_type="cnn", kl_regularizer=False,
               proj_key=_linear(wk, bk, dev), proj_value=_linear(wv, bv, dev), sigma_0=None, mu_0=None,
               sticky_memories=case.sticky, sigmas=None, tau=case.tau, d_model=case.dm)


@pytest.mark.parametrize("case", [CASES[0], CASES[1], CASES[2], CASES[7]], ids=lambda c: c.name)
def test_dropin_module_reproduces_reference_run(case):
    """Same call sequence and torch.manual_seed discipline as make_goldens.py used on the reference."""
    dev = torch.device("cuda:0")
    g = load_golden(case)
    ks, qs, ws = case_inputs(case)
    mods = [_module(case, ws, l, dev) for l in range(case.n_layers)]
    for m in mods:
        assert m.B_past is None
    with torch.no_grad():
        for c in range(len(case.chunk_T)):
            k = torch.from_numpy(ks[c]).unsqueeze(0).to(dev)
            for l, m in enumerate(mods):
                tag = f"c{c}_l{l}"
                torch.manual_seed(call_seed(case, c, l))
                m.length = m.target_len = k.size(1)               # as the hook does, Qformer.py:218-219
                out = m(k, torch.from_numpy(qs[l]).unsqueeze(0).to(dev), new_doc=(c in case.new_doc_at), layer_n=l)
                assert out.shape == (1, case.Q, case.dm)
                # same position of torch's CPU generator afterwards as the reference
                np.testing.assert_array_equal(torch.rand(1, dtype=torch.float64).numpy(), g[tag + "_next_u"])
                np.testing.assert_allclose(out[0].cpu().numpy(), g[tag + "_ctx"], rtol=0, atol=CTX_TOL)
                B = m.B_past
                assert B.shape == (1, case.N, case.d)
                if case.store_full_B:
                    np.testing.assert_allclose(B[0].cpu().numpy(), g[tag + "_B"], rtol=0, atol=2e-5)
                if tag + "_bins" in g:
                    np.testing.assert_array_equal(m._engine.last_draw(0)[0], g[tag + "_bins"])


def test_hook_call_and_merge_rules():
    from infinite_video_amd.qformer_hook import LongTermMemoryHook
    dev = torch.device("cuda:0")
    case = CASES[0]
    ks, qs, ws = case_inputs(case)
    wk, bk, wv, bv = ws[0]
    key, value = _linear(wk, bk, dev), _linear(wv, bv, dev)
    k = torch.from_numpy(ks[0]).unsqueeze(0).to(dev)
    q = torch.from_numpy(qs[0]).unsqueeze(0).to(dev)
    short = torch.randn(1, case.Q, case.dm, device=dev)
    cfg = types.SimpleNamespace(num_basis=case.N, encoder_width=case.d, sticky=True, sigmas=None, tau=.75, alpha=.75)
    hook = LongTermMemoryHook(cfg, key, value, case.H, case.dh)
    # image Q-former path: no position embedding -> hook inert (Qformer.py:216,303)
    assert hook.long_term(k, q, None, 0, True) == 0
    assert hook.merge(short, 0, None) is short
    a_long = hook.long_term(k, q, object(), 0, True)
    g = load_golden(case)
    np.testing.assert_allclose(a_long[0].cpu().numpy(), g["c0_l0_ctx"], rtol=0, atol=CTX_TOL)
    # the hook sets length = target_len = p (Qformer.py:218-219); forward then resets length to the
    # frame count (reference :292)
    assert hook.long_term_attention.target_len == k.shape[1] and hook.long_term_attention.length == case.chunk_T[0]
    assert not a_long.requires_grad
    merged = hook.merge(short, a_long, object())
    torch.testing.assert_close(merged, .75 * short + .25 * a_long)
    # alpha == 1.0: the reference never calls the op (Qformer.py:220-223)
    cfg1 = types.SimpleNamespace(num_basis=case.N, encoder_width=case.d, sticky=True, sigmas=None, tau=.75, alpha=1.0)
    hook1 = LongTermMemoryHook(cfg1, key, value, case.H, case.dh)
    assert hook1.long_term(k, q, object(), 0, True) == 0
    assert hook1.long_term_attention.B_past is None
    torch.testing.assert_close(hook1.merge(short, 0, object()), short)


def test_install_redirects_reference_import_site():
    import importlib
    import sys
    from infinite_video_amd import qformer_hook
    from infinite_video_amd.long_term_attention_gibbs import LongTermAttention
    qformer_hook.install("VL")
    try:
        mod = importlib.import_module("InfVideoLLaMA.models.long_term_attention_gibbs")
        assert mod.LongTermAttention is LongTermAttention
    finally:
        sys.modules.pop("InfVideoLLaMA.models.long_term_attention_gibbs", None)


def test_memory_state_roundtrip_through_cpu():
    """memory_state()/load_memory_state(): a second module continues the document exactly where the first stopped."""
    dev = torch.device("cuda:0")
    case = CASES[0]
    ks, qs, ws = case_inputs(case)
    a, b = _module(case, ws, 0, dev), _module(case, ws, 0, dev)
    q = torch.from_numpy(qs[0]).unsqueeze(0).to(dev)
    assert a.memory_state() is None
    with torch.no_grad():
        for c in range(3):
            torch.manual_seed(call_seed(case, c, 0))
            a(torch.from_numpy(ks[c]).unsqueeze(0).to(dev), q, new_doc=(c == 0), layer_n=0)
        state = a.memory_state()
        assert state["B_past"].device.type == "cpu" and state["B_past"].shape == (case.N, case.d)
        b.load_memory_state(state, dev)
        k3 = torch.from_numpy(ks[3]).unsqueeze(0).to(dev)
        torch.manual_seed(call_seed(case, 3, 0)); ya = a(k3, q, new_doc=False, layer_n=0)
        torch.manual_seed(call_seed(case, 3, 0)); yb = b(k3, q, new_doc=False, layer_n=0)
    np.testing.assert_array_equal(a._engine.last_draw(0)[0], b._engine.last_draw(0)[0])
    np.testing.assert_allclose(ya.cpu().numpy(), yb.cpu().numpy(), rtol=0, atol=2e-5)
    g = load_golden(case)
    np.testing.assert_allclose(yb[0].cpu().numpy(), g["c3_l0_ctx"], rtol=0, atol=CTX_TOL)


def test_half_precision_producer_and_weights():
    """VideoChat2 feeds the op from an fp16 autocast region (videochat2_it_mistral.py:187): fp16 frame tokens, queries
    and key/value weights must be accepted, computed in fp32 on the values they hold, and returned in the query's
    dtype -- compared with the same module fed the identical (fp16-rounded) numbers as fp32."""
    case = CASES[0]
    dev = torch.device("cuda:0")
    ks, qs, ws = case_inputs(case)
    r16 = lambda a: torch.from_numpy(a).half()                   # the values an fp16 producer would hold
    ws16 = [tuple(r16(w).float().numpy() for w in ws[l]) for l in range(case.n_layers)]
    ref = _module(case, ws16, 0, dev)                            # fp32 modules holding the fp16-rounded weights
    low = _module(case, ws16, 0, dev)
    low.proj_key.half(); low.proj_value.half()
    for c in range(3):
        k16, q16 = r16(ks[c]).unsqueeze(0).to(dev), r16(qs[0]).unsqueeze(0).to(dev)
        torch.manual_seed(call_seed(case, c, 0))
        out16 = low(k16, q16, new_doc=(c == 0), layer_n=0)
        torch.manual_seed(call_seed(case, c, 0))
        out32 = ref(k16.float(), q16.float(), new_doc=(c == 0), layer_n=0)
        assert out16.dtype == torch.float16 and out32.dtype == torch.float32
        np.testing.assert_allclose(out16.float().cpu().numpy(), out32.cpu().numpy(), atol=2e-3, rtol=2e-3)   # fp16 output rounding
        np.testing.assert_array_equal(low._engine.last_draw(0)[0], ref._engine.last_draw(0)[0])
