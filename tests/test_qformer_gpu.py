"""HIP video Q-former path (infv_vqf_*) against the goldens captured from the REAL reference BertEncoder and
against the CPU oracle.  Tolerance: the north star asks 1e-3 fp32; these tests hold 2e-4."""
import math

import numpy as np
import pytest
import torch

from tests.golden.qformer_cases import QF_CASES, chunk_seed, chunk_uniforms, load_qf_golden, qf_inputs

pytestmark = pytest.mark.gpu
ATOL = 2e-4


def make_model(case, weights, dev):
    from infinite_video_amd.video_qformer import InfVideoEncoder
    m = InfVideoEncoder(num_video_query_token=case.n_query, hidden_size=case.hidden, llama_hidden=case.proj_out,
                        sticky=case.sticky, num_basis=case.N, tau=case.tau, alpha=case.alpha,
                        num_hidden_layers=case.n_layers)
    m.load_reference_state_dict(weights)
    return m.to(dev)


def torch_short_attention(frames, xq, wk, bk, wv, bv, H=12):
    """Plain PyTorch fp32 reference of the short-term cross-attention (Qformer.py:225-301 with zero masks)."""
    dh = xq.shape[1] // H
    K = torch.nn.functional.linear(frames, wk, bk)
    V = torch.nn.functional.linear(frames, wv, bv)
    heads = lambda x: x.reshape(x.shape[0], H, dh).permute(1, 0, 2)
    s = torch.matmul(heads(xq), heads(K).transpose(-1, -2)) / math.sqrt(dh)
    return torch.matmul(torch.softmax(s, -1), heads(V)).permute(1, 0, 2).reshape(xq.shape[0], -1)


@pytest.mark.parametrize("T,gain", [(8, 1.0), (5, 1.0), (1, 4.0), (64, 6.0), (256, 1.0), (256, 10.0)])
def test_short_attention_matches_torch(T, gain):
    import ctypes as C
    from infinite_video_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(100 + T)
    frames = torch.randn(T * 32, 768, generator=g)
    xq = torch.randn(32, 768, generator=g) * gain
    wk, wv = torch.randn(768, 768, generator=g) * 0.02, torch.randn(768, 768, generator=g) * 0.02
    bk, bv = torch.randn(768, generator=g) * 0.02, torch.randn(768, generator=g) * 0.02
    along = torch.randn(32, 768, generator=g)
    want_short = torch_short_attention(frames.double(), xq.double(), wk.double(), bk.double(), wv.double(), bv.double())
    lib = _lib.load()
    for alpha, use_long, exact in ((0.9, True, True), (1.0, False, True), (0.9, True, False), (1.0, False, False)):
        cfg = _lib.VqfConfig(2, 12, 768, 3072, 768, 32, 32, 4096, 512, alpha, 1e-12)
        h = C.c_void_p()
        _lib.check(lib.infv_vqf_create(C.byref(cfg), C.byref(h)))
        _lib.check(lib.infv_vqf_set_precision(h, int(exact)))
        d = lambda t: t.to(dev).contiguous()
        fr, q, a = d(frames), d(xq), d(along)
        dwk, dbk, dwv, dbv = d(wk), d(bk), d(wv), d(bv)
        key, val = _lib.Linear(dwk.data_ptr(), dbk.data_ptr()), _lib.Linear(dwv.data_ptr(), dbv.data_ptr())
        out = torch.empty(32, 768, device=dev)
        _lib.check(lib.infv_vqf_short_attention(h, C.c_void_p(fr.data_ptr()), T * 32, C.c_void_p(q.data_ptr()),
                                                C.byref(key), C.byref(val),
                                                C.c_void_p(a.data_ptr() if use_long else 0),
                                                C.c_void_p(out.data_ptr()),
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        want = alpha * want_short + (1 - alpha) * along.double() if use_long else want_short
        # exact fp32 MFMA: fp32 round-off only; split-bf16 (default): ~1e-5 relative, a tenth of the 1e-3 budget
        tol = dict(atol=2e-5, rtol=1e-4) if exact else dict(atol=1e-4, rtol=1e-3)
        np.testing.assert_allclose(out.cpu().numpy(), want.float().numpy(), **tol)
        _lib.check(lib.infv_vqf_destroy(h))


@pytest.mark.parametrize("case", QF_CASES, ids=lambda c: c.name)
def test_encode_chunks_match_reference_goldens(case):
    dev = torch.device("cuda:0")
    frames, weights = qf_inputs(case)
    g = load_qf_golden(case)
    m = make_model(case, weights, dev)
    for c in range(len(case.chunk_T)):
        k = torch.from_numpy(frames[c]).unsqueeze(0).to(dev)
        hid, llama = m.encode_frames(k, new_video=(c == 0), u=torch.from_numpy(chunk_uniforms(case, c)))
        torch.cuda.synchronize()
        np.testing.assert_allclose(hid[0].cpu().numpy(), g[f"c{c}_hidden"], atol=ATOL, err_msg=f"hidden c{c}")
        np.testing.assert_allclose(llama[0].cpu().numpy(), g[f"c{c}_llama"], atol=ATOL, err_msg=f"llama c{c}")
        if case.alpha != 1.0:
            for l, ltm in enumerate(m.video_Qformer.ltm_modules):
                Bsum = ltm.B_past[0].double().sum(1).cpu().numpy()
                np.testing.assert_allclose(Bsum, g[f"c{c}_l{l}_Bsum"], atol=5e-4, err_msg=f"B c{c} l{l}")


def test_encode_video_consumes_global_rng_like_the_reference():
    case = QF_CASES[0]
    dev = torch.device("cuda:0")
    frames, weights = qf_inputs(case)
    g = load_qf_golden(case)
    m = make_model(case, weights, dev)
    for c, T in enumerate(case.chunk_T):
        blocks = torch.from_numpy(frames[c]).reshape(T, case.P, case.hidden).to(dev)
        m.short_memory_buffer = list(blocks)
        torch.manual_seed(chunk_seed(case, c))
        inputs_llama, atts = m.encode_video(new_video=(c == 0))
        nxt = torch.rand(1, dtype=torch.float64).numpy()
        assert nxt == g[f"c{c}_next_u"], "generator position after the chunk differs from the reference"
        assert atts.shape == (1, case.n_query) and atts.dtype == torch.long and bool((atts == 1).all())
        np.testing.assert_allclose(inputs_llama[0].cpu().numpy(), g[f"c{c}_llama"], atol=ATOL)


def test_loop_counterpart_ragged_split_and_mean():
    from infinite_video_amd.video_qformer import encode_long_video
    case = QF_CASES[0]                                   # chunk_T = [8, 8, 5, ...]: 21 frames split at 8
    dev = torch.device("cuda:0")
    frames, weights = qf_inputs(case)
    g = load_qf_golden(case)
    m = make_model(case, weights, dev)
    video = torch.cat([torch.from_numpy(f).reshape(-1, case.P, case.hidden) for f in frames[:3]]).to(dev)
    mean, embs = encode_long_video(m, video, 8, u_of_chunk=lambda i: torch.from_numpy(chunk_uniforms(case, i)))
    assert len(embs) == 3
    want = np.mean(np.stack([g[f"c{c}_llama"] for c in range(3)]), 0)
    np.testing.assert_allclose(mean[0].cpu().numpy(), want, atol=ATOL)


def test_frame_cap_drops_oldest_frames():
    case = QF_CASES[1]                                   # alpha = 1: no memory, cheap
    dev = torch.device("cuda:0")
    _, weights = qf_inputs(case)
    m = make_model(case, weights, dev)
    gen = torch.Generator().manual_seed(5)
    blocks = torch.randn(1030, case.P, case.hidden, generator=gen).to(dev)
    m.short_memory_buffer = list(blocks)
    full, _ = m.encode_video(new_video=True)
    assert m.n_position == 32 and len(m.short_memory_buffer) == 1024
    m.short_memory_buffer = list(blocks[6:])
    newest, _ = m.encode_video(new_video=True)
    np.testing.assert_array_equal(full.cpu().numpy(), newest.cpu().numpy())


def test_errors():
    case = QF_CASES[1]
    _, weights = qf_inputs(case)
    m = make_model(case, weights, torch.device("cuda:0"))
    with pytest.raises(RuntimeError):
        m.encode_frames(torch.zeros(1, 64, 768), new_video=True)            # CPU tensor: no fallback
    with pytest.raises(ValueError):
        m.encode_frames(torch.zeros(2, 64, 768, device="cuda:0"), new_video=True)
    with pytest.raises(ValueError):
        m.encode_frames(torch.zeros(1, 65, 768, device="cuda:0"), new_video=True)


def test_memory_file_roundtrip_continues_the_document(tmp_path):
    """Consolidate two chunks, write the memory file, load it into a fresh model: chunk 3 matches the golden."""
    from infinite_video_amd.memory_io import load_memory, save_memory
    case = QF_CASES[0]
    dev = torch.device("cuda:0")
    frames, weights = qf_inputs(case)
    g = load_qf_golden(case)
    a = make_model(case, weights, dev)
    with pytest.raises(RuntimeError):
        save_memory(str(tmp_path / "empty.safetensors"), a.video_Qformer.ltm_modules)
    for c in range(2):
        a.encode_frames(torch.from_numpy(frames[c]).unsqueeze(0).to(dev), new_video=(c == 0),
                        u=torch.from_numpy(chunk_uniforms(case, c)))
    path = str(tmp_path / "video.safetensors")
    save_memory(path, a.video_Qformer.ltm_modules, user={"video": "qf_small", "chunks": 2})
    b = make_model(case, weights, dev)
    assert load_memory(path, b.video_Qformer.ltm_modules, dev) == {"video": "qf_small", "chunks": "2"}
    for c in (2, 3):
        _, llama = b.encode_frames(torch.from_numpy(frames[c]).unsqueeze(0).to(dev), new_video=False,
                                   u=torch.from_numpy(chunk_uniforms(case, c)))
        np.testing.assert_allclose(llama[0].cpu().numpy(), g[f"c{c}_llama"], atol=ATOL)


@pytest.mark.parametrize("case", [c for c in QF_CASES if len(set(c.chunk_T)) == 1], ids=lambda c: c.name)
def test_layer_major_whole_video_matches_goldens_and_per_chunk(case):
    """infv_vqf_encode_video (layer-major, LTM layer 0 on the whole-video fast path) == per-chunk encode == reference."""
    dev = torch.device("cuda:0")
    frames, weights = qf_inputs(case)
    g = load_qf_golden(case)
    m = make_model(case, weights, dev)
    Cn = len(case.chunk_T)
    k = torch.from_numpy(np.stack(frames)).to(dev)
    u = torch.from_numpy(np.stack([chunk_uniforms(case, c) for c in range(Cn)]))
    llama, mean, hidden = m.encode_frames_batch(k, new_video=True, u=u, want_hidden=True)
    torch.cuda.synchronize()
    for c in range(Cn):
        np.testing.assert_allclose(hidden[c].cpu().numpy(), g[f"c{c}_hidden"], atol=ATOL, err_msg=f"hidden c{c}")
        np.testing.assert_allclose(llama[c].cpu().numpy(), g[f"c{c}_llama"], atol=ATOL, err_msg=f"llama c{c}")
    want = np.mean(np.stack([g[f"c{c}_llama"] for c in range(Cn)]), 0)
    np.testing.assert_allclose(mean[0].cpu().numpy(), want, atol=ATOL)
    if case.alpha != 1.0:
        for l, ltm in enumerate(m.video_Qformer.ltm_modules):
            Bsum = ltm.B_past[0].double().sum(1).cpu().numpy()
            np.testing.assert_allclose(Bsum, g[f"c{Cn - 1}_l{l}_Bsum"], atol=5e-4, err_msg=f"final B l{l}")


def test_layer_major_path_with_ragged_tail_and_global_rng():
    from infinite_video_amd.video_qformer import encode_long_video
    case = QF_CASES[0]                                   # 8, 8, 5: two full chunks batched + ragged tail per chunk
    dev = torch.device("cuda:0")
    frames, weights = qf_inputs(case)
    g = load_qf_golden(case)
    m = make_model(case, weights, dev)
    video = torch.cat([torch.from_numpy(f).reshape(-1, case.P, case.hidden) for f in frames[:3]]).to(dev)
    mean, embs = encode_long_video(m, video, 8, u_of_chunk=lambda i: torch.from_numpy(chunk_uniforms(case, i)),
                                   batched=True)
    assert len(embs) == 3
    for c in range(3):
        np.testing.assert_allclose(embs[c][0].cpu().numpy(), g[f"c{c}_llama"], atol=ATOL, err_msg=f"c{c}")
    np.testing.assert_allclose(mean[0].cpu().numpy(), np.mean(np.stack([g[f"c{c}_llama"] for c in range(3)]), 0), atol=ATOL)


def test_cached_layer0_prefix_follows_weight_updates():
    """The chunk-independent prefix of layer 0 is reused between calls; an in-place weight update must invalidate it."""
    case = QF_CASES[1]                                   # alpha = 1: no memory, outputs depend on weights + frames only
    dev = torch.device("cuda:0")
    frames, weights = qf_inputs(case)
    m = make_model(case, weights, dev)
    k = torch.from_numpy(frames[0]).unsqueeze(0).to(dev)
    a1 = m.encode_frames(k, new_video=True)[1].clone()
    a2 = m.encode_frames(k, new_video=True)[1].clone()          # second call: cached prefix
    np.testing.assert_array_equal(a1.cpu().numpy(), a2.cpu().numpy())
    with torch.no_grad():
        m.video_Qformer.bert.encoder.layer[0].attention.self.query.weight.mul_(1.5)
    b = m.encode_frames(k, new_video=True)[1].clone()
    fresh = make_model(case, weights, dev)
    with torch.no_grad():
        fresh.video_Qformer.bert.encoder.layer[0].attention.self.query.weight.mul_(1.5)
    want = fresh.encode_frames(k, new_video=True)[1]
    np.testing.assert_array_equal(b.cpu().numpy(), want.cpu().numpy())
    assert float((a1 - b).abs().max()) > 1e-4


def test_sharded_entry_point_single_rank():
    """world = 1 (no process group): the sharded entry point equals the layer-major path and returns the memories."""
    from infinite_video_amd.video_qformer import encode_long_video_sharded
    case = QF_CASES[3]                                    # peaked, 3 equal chunks
    dev = torch.device("cuda:0")
    frames, weights = qf_inputs(case)
    g = load_qf_golden(case)
    m = make_model(case, weights, dev)
    Cn = len(case.chunk_T)
    k = torch.from_numpy(np.stack(frames)).to(dev)
    u = torch.from_numpy(np.stack([chunk_uniforms(case, c) for c in range(Cn)]))
    mean, per_rank, llama = encode_long_video_sharded(m, k, u)
    want = np.mean(np.stack([g[f"c{c}_llama"] for c in range(Cn)]), 0)
    np.testing.assert_allclose(mean[0].cpu().numpy(), want, atol=ATOL)
    assert len(per_rank) == 1 and len(per_rank[0]) == 2 * case.n_layers
    np.testing.assert_allclose(per_rank[0][0].double().sum(1).cpu().numpy(), g[f"c{Cn - 1}_l0_Bsum"], atol=5e-4)


def test_short_attention_rows_sum_to_one_at_headline_size():
    """Size-independent property at T=256 (8192 tokens): with a zero value projection the short-term context is the
    value bias exactly (softmax rows sum to one), for both arithmetic modes."""
    import ctypes as C
    from infinite_video_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(321)
    frames = torch.randn(256 * 32, 768, generator=g).to(dev)
    xq = (torch.randn(32, 768, generator=g) * 3).to(dev)
    wk = (torch.randn(768, 768, generator=g) * 0.02).to(dev)
    bk = torch.zeros(768, device=dev)
    wv = torch.zeros(768, 768, device=dev)
    bv = torch.randn(768, generator=g).to(dev)
    lib = _lib.load()
    for exact in (1, 0):
        cfg = _lib.VqfConfig(2, 12, 768, 3072, 768, 32, 32, 4096, 512, 1.0, 1e-12)
        h = C.c_void_p()
        _lib.check(lib.infv_vqf_create(C.byref(cfg), C.byref(h)))
        _lib.check(lib.infv_vqf_set_precision(h, exact))
        key, val = _lib.Linear(wk.data_ptr(), bk.data_ptr()), _lib.Linear(wv.data_ptr(), bv.data_ptr())
        out = torch.empty(32, 768, device=dev)
        _lib.check(lib.infv_vqf_short_attention(h, C.c_void_p(frames.data_ptr()), 256 * 32, C.c_void_p(xq.data_ptr()),
                                                C.byref(key), C.byref(val), C.c_void_p(0), C.c_void_p(out.data_ptr()),
                                                C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        np.testing.assert_allclose(out.cpu().numpy(), bv.unsqueeze(0).expand(32, -1).cpu().numpy(), atol=1e-6)
        _lib.check(lib.infv_vqf_destroy(h))


def test_encode_video_from_the_producer_layout_equals_the_list_path():
    """ShortMemoryBuffer (tokens kept as one [T, P, d] block, infinityqa.py:251-278 counterpart) feeds encode_video the
    same frames as the reference's list of per-frame tensors, including the frame cap."""
    from infinite_video_amd.video_qformer import ShortMemoryBuffer
    case = QF_CASES[0]
    dev = torch.device("cuda:0")
    frames, weights = qf_inputs(case)
    a, b = make_model(case, weights, dev), make_model(case, weights, dev)
    P, d = 32, 768
    buf = ShortMemoryBuffer(P, d, capacity_frames=64, device=dev)
    for c, fr in enumerate(frames[:3]):
        tok = torch.from_numpy(fr).to(dev).reshape(-1, P, d)
        torch.manual_seed(chunk_seed(case, c))
        a.short_memory_buffer = list(tok)
        ya, _ = a.encode_video(new_video=(c == 0))
        torch.manual_seed(chunk_seed(case, c))
        b.short_memory_buffer = buf.replace(tok)
        yb, _ = b.encode_video(new_video=(c == 0))
        assert torch.equal(ya, yb)
        assert b.n_position == a.n_position


def test_single_token_pass_equals_separate_pooling_and_per_layer_split(monkeypatch):
    """One pass over the frame tokens (split + transpose + frame means, shared by both layers and their memories) gives
    the same bits as the round-1 arrangement (infv_ltm_pool + one split pass per layer): per chunk and layer-major."""
    dev = torch.device("cuda:0")
    case = next(c for c in QF_CASES if c.alpha != 1.0 and len(set(c.chunk_T)) == 1 and c.chunk_T[0] % 2 == 0)
    frames, weights = qf_inputs(case)
    Cn = len(case.chunk_T)
    k = torch.from_numpy(np.stack(frames)).to(dev)
    u = torch.from_numpy(np.stack([chunk_uniforms(case, c) for c in range(Cn)]))
    outs = {}
    for fuse in ("1", "0", "nocache"):
        # (both knobs are read when the handle is created)  "nocache": single token pass per chunk call, but the whole video's
        # split copies do not fit the budget, so the layer-major schedule splits per sub-batch and pools from the tokens
        monkeypatch.setenv("INFV_VQF_FUSE", "0" if fuse == "0" else "1")
        monkeypatch.setenv("INFV_VQF_SPLIT_CACHE_GB", "0" if fuse == "nocache" else "64")
        m = make_model(case, weights, dev)
        per_chunk = [m.encode_frames(k[c:c + 1], new_video=(c == 0), u=u[c])[0].clone() for c in range(Cn)]
        llama, mean, hidden = m.encode_frames_batch(k, new_video=True, u=u, want_hidden=True)
        torch.cuda.synchronize()
        outs[fuse] = (torch.stack(per_chunk), llama.clone(), hidden.clone())
    for other in ("0", "nocache"):
        for a, b, what in zip(outs["1"], outs[other], ("per chunk", "layer-major llama", "layer-major hidden")):
            assert torch.equal(a, b), f"{other}: {what}"
