"""VideoChat2 binding (BASELINE configs[4]): the HIP LTM operator and short-term cross-attention bound into the
12-layer VideoChat2 Q-former, against the goldens captured from the REAL reference encoder and the CPU oracle.
8 chunks of 16 frames, 6 LTM layers, 96 query tokens + 8 text tokens.  Needs a real MI355X: run with ``-m gpu``."""
import numpy as np
import pytest
import torch

from oracle.videochat2_oracle import VideoChat2Oracle
from tests.golden.vc_cases import VC_CASE, chunk_seed, chunk_uniforms, load_vc_golden, vc_inputs

pytestmark = pytest.mark.gpu
ATOL = 2e-4            # north star 1e-3 fp32


def make_model(case, weights, dev):
    from infinite_video_amd.videochat2_qformer import VideoChat2Encoder
    m = VideoChat2Encoder(num_query_token=32, extra_num_query_token=case.n_query - 32, vision_width=case.enc_width,
                          llm_hidden=case.proj_out, num_basis=case.N, sticky=case.sticky, tau=case.tau, alpha=case.alpha,
                          num_hidden_layers=case.n_layers, cross_attention_freq=case.cross_freq)
    m.load_reference_state_dict(weights)
    return m.to(dev).eval()


def test_videochat2_chunks_match_the_reference_goldens():
    case = VC_CASE
    dev = torch.device("cuda:0")
    g = load_vc_golden(case)
    frames, h0, weights = vc_inputs(case)
    m = make_model(case, weights, dev)
    assert len(m.qformer.ltm_modules) == 6
    orc = VideoChat2Oracle(weights, case.N, case.tau, case.alpha, case.sticky, case.n_layers, case.cross_freq, 12,
                           case.n_query, case.P)
    hin = torch.from_numpy(h0).unsqueeze(0).to(dev)
    T = case.max_int
    embs = []
    for c in range(case.num_samples):
        blk = torch.from_numpy(frames[c * T:(c + 1) * T]).to(dev).reshape(1, -1, case.enc_width)
        torch.manual_seed(chunk_seed(case, c))                   # the six LTM calls draw like the reference's
        emb, hid = m.encode_tokens(blk, new_video=(c == 0), hidden_in=hin)
        assert float(torch.rand(1, dtype=torch.float64)) == float(g[f"c{c}_next_u"][0])     # generator left where the reference leaves it
        embs.append(emb)
        np.testing.assert_allclose(emb[0].cpu().numpy(), g[f"c{c}_mistral"], rtol=0, atol=ATOL, err_msg=f"chunk {c}")
        if f"c{c}_hidden" in g.files:
            np.testing.assert_allclose(hid[0].cpu().numpy(), g[f"c{c}_hidden"], rtol=0, atol=ATOL)
        ref_h, ref_e = orc.encode_chunk(frames[c * T:(c + 1) * T].reshape(-1, case.enc_width), h0, c == 0, chunk_uniforms(case, c))
        np.testing.assert_allclose(emb[0].cpu().numpy(), ref_e, rtol=0, atol=ATOL)
        for j, mod in enumerate(m.qformer.ltm_modules):
            B = mod.B_past[0].cpu().numpy()
            np.testing.assert_allclose(B.astype(np.float64).sum(1), g[f"c{c}_l{2 * j}_Bsum"], rtol=0, atol=1e-3)
            if c > 0:
                np.testing.assert_array_equal(mod._engine.last_draw(0)[0], orc.ltm[2 * j].last_bins)
    mean = torch.mean(torch.stack(embs), dim=0, keepdim=True).squeeze(0)
    np.testing.assert_allclose(mean[0].cpu().numpy(), g["mean_mistral"], rtol=0, atol=ATOL)


def test_videochat2_eval_loop_counterpart():
    """infer_*_inf (run_nextqa_mistral.py:141-152): torch.chunk over frames, new_video on the first chunk, mean."""
    from infinite_video_amd.videochat2_qformer import encode_long_video_vc
    case = VC_CASE
    dev = torch.device("cuda:0")
    g = load_vc_golden(case)
    frames, h0, weights = vc_inputs(case)
    m = make_model(case, weights, dev)
    hin = torch.from_numpy(h0).unsqueeze(0).to(dev)

    class Seeder:                                                # re-seed before every chunk like the golden run
        def __init__(self):
            self.c = 0

    video = torch.from_numpy(frames).to(dev)
    embs = []
    new_video = True
    for c, blk in enumerate(torch.chunk(video, case.num_samples, dim=0)):
        torch.manual_seed(chunk_seed(case, c))
        emb, _ = m.encode_tokens(blk.reshape(1, -1, case.enc_width), new_video=new_video, hidden_in=hin)
        embs.append(emb)
        new_video = False
    mean_manual = torch.mean(torch.stack(embs), dim=0, keepdim=True).squeeze(0)
    np.testing.assert_allclose(mean_manual[0].cpu().numpy(), g["mean_mistral"], rtol=0, atol=ATOL)
    # the packaged loop on another chunking: 120 frames in 5 chunks of 24, and a ragged one (100 frames -> 13, ..., 9:
    # token counts that are no multiple of 32 take the stock-PyTorch cross-attention), both against the CPU oracle
    for F_, ns in ((120, 5), (100, 8)):
        m2 = make_model(case, weights, dev)
        orc = VideoChat2Oracle(weights, case.N, case.tau, case.alpha, case.sticky, case.n_layers, case.cross_freq, 12,
                               case.n_query, case.P)
        us = {}

        def u_of_chunk(c):
            torch.manual_seed(777 + c)
            us[c] = np.stack([torch.rand(512, dtype=torch.float64).numpy() for _ in range(12)])[::2]   # draws 0, 2, 4, ... of the stream
            return us[c]

        want_mean, want = orc.encode_long_video(frames[:F_], h0, ns, u_of_chunk)
        embs = []
        for c, blk in enumerate(torch.chunk(video[:F_], ns, dim=0)):
            torch.manual_seed(777 + c)
            embs.append(m2.encode_tokens(blk.reshape(1, -1, case.enc_width), new_video=(c == 0), hidden_in=hin)[0])
        assert len(embs) == len(want)
        for c in range(len(want)):
            np.testing.assert_allclose(embs[c][0].cpu().numpy(), want[c], rtol=0, atol=ATOL, err_msg=f"F={F_} chunk {c}")
    m3 = make_model(case, weights, dev)
    torch.manual_seed(1)
    mean, per_chunk = encode_long_video_vc(m3, video[:120], 5, hidden_in=hin)
    assert len(per_chunk) == 5 and mean.shape == (1, case.n_query, case.proj_out)
    torch.testing.assert_close(mean, torch.stack(per_chunk).mean(0), rtol=1e-6, atol=1e-6)
    assert m3.qformer.ltm_modules[0]._engine.has_memory


def test_videochat2_alpha_one_bypasses_the_memory():
    """alpha == 1.0: the reference skips the LTM call (Qformer.py:218-221) -- no memory is built, no uniforms drawn."""
    case = VC_CASE
    dev = torch.device("cuda:0")
    frames, h0, weights = vc_inputs(case)
    from infinite_video_amd.videochat2_qformer import VideoChat2Encoder
    m = VideoChat2Encoder(32, 64, case.enc_width, case.proj_out, case.N, True, case.tau, alpha=1.0)
    m.load_reference_state_dict(weights)
    m = m.to(dev).eval()
    hin = torch.from_numpy(h0).unsqueeze(0).to(dev)
    blk = torch.from_numpy(frames[:16]).to(dev).reshape(1, -1, case.enc_width)
    torch.manual_seed(5)
    before = torch.get_rng_state()
    emb, _ = m.encode_tokens(blk, new_video=True, hidden_in=hin)
    emb2, _ = m.encode_tokens(blk, new_video=False, hidden_in=hin)
    assert torch.equal(torch.get_rng_state(), before)
    assert torch.equal(emb, emb2) and bool(torch.isfinite(emb).all())
    assert all(mod._engine is None or not mod._engine.has_memory for mod in m.qformer.ltm_modules)
