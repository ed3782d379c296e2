"""The Q-former restatement (oracle/qformer_oracle.py) against goldens captured from the REAL reference
BertEncoder + LongTermAttention (tests/golden/make_qformer_goldens.py).  CPU only."""
import numpy as np
import pytest

from oracle.qformer_oracle import VideoQformerOracle, frame_cap
from tests.golden.qformer_cases import QF_CASES, chunk_uniforms, load_qf_golden, qf_inputs


def make_oracle(case, weights):
    return VideoQformerOracle(weights, case.N, case.tau, case.alpha, case.sticky, case.n_layers)


@pytest.mark.parametrize("case", QF_CASES, ids=lambda c: c.name)
def test_oracle_matches_reference_encoder(case):
    frames, weights = qf_inputs(case)
    g = load_qf_golden(case)
    orc = make_oracle(case, weights)
    np.testing.assert_allclose(orc.embed().numpy(), g["h0"], atol=2e-6)
    for c in range(len(case.chunk_T)):
        hid, llama = orc.encode_chunk(frames[c], new_video=(c == 0), u=chunk_uniforms(case, c))
        for l in range(case.n_layers):
            np.testing.assert_allclose(orc.taps[f"l{l}_xq"], g[f"c{c}_l{l}_xq"], atol=2e-5, err_msg=f"xq c{c} l{l}")
            if case.alpha != 1.0:
                np.testing.assert_allclose(orc.taps[f"l{l}_along"], g[f"c{c}_l{l}_along"], atol=2e-5,
                                           err_msg=f"a_long c{c} l{l}")
                Bsum = orc.ltm[l].B_past.astype(np.float64).sum(1)
                np.testing.assert_allclose(Bsum, g[f"c{c}_l{l}_Bsum"], atol=2e-4, err_msg=f"B c{c} l{l}")
            np.testing.assert_allclose(orc.taps[f"l{l}_xctx"], g[f"c{c}_l{l}_xctx"], atol=2e-5, err_msg=f"xctx c{c} l{l}")
        np.testing.assert_allclose(hid, g[f"c{c}_hidden"], atol=5e-5, err_msg=f"hidden c{c}")
        np.testing.assert_allclose(llama, g[f"c{c}_llama"], atol=5e-5, err_msg=f"llama c{c}")


def test_alpha_one_bypasses_ltm():
    case = [c for c in QF_CASES if c.alpha == 1.0][0]
    g = load_qf_golden(case)
    assert not any("along" in k or "Bsum" in k for k in g.files)      # the reference never called the op


def test_frame_cap_rule():
    # infinityqa.py:285-288,306-307: n_position = min(32, ceil(sqrt(n))); keep the newest n_position^2 frames
    assert [frame_cap(n) for n in (1, 2, 8, 255, 256, 1024, 1025, 1030, 5000)] == [1, 2, 8, 255, 256, 1024, 1024, 1024, 1024]


def test_loop_counterpart_mean_and_ragged_split():
    case = QF_CASES[0]
    frames, weights = qf_inputs(case)
    g = load_qf_golden(case)
    orc = make_oracle(case, weights)
    # chunk_T = [8, 8, 5, 8] is not a torch.split pattern; use the first two full chunks + the ragged third as
    # the tail of a 21-frame video split at max_int = 8 (run_inference_inf_video_llama_nextqa.py:181-194,228)
    video = np.concatenate([f.reshape(-1, case.P, case.hidden) for f in frames[:3]], 0)
    assert video.shape[0] == 21
    mean, embs = orc.encode_long_video(video, 8, lambda i: chunk_uniforms(case, i))
    assert len(embs) == 3
    for c in range(3):
        np.testing.assert_allclose(embs[c], g[f"c{c}_llama"], atol=5e-5)
    np.testing.assert_allclose(mean, np.mean(np.stack([g[f"c{c}_llama"] for c in range(3)]), 0), atol=5e-5)
