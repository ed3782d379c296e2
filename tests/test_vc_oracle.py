"""The CPU restatement of the VideoChat2 Q-former path (oracle/videochat2_oracle.py) against the golden vectors
captured from the REAL reference encoder (tests/golden/make_vc_goldens.py).  Runs anywhere (no GPU)."""
import numpy as np

from oracle.videochat2_oracle import VideoChat2Oracle
from tests.golden.vc_cases import VC_CASE, chunk_uniforms, load_vc_golden, vc_inputs


def test_videochat2_oracle_reproduces_the_reference_encoder():
    case = VC_CASE
    g = load_vc_golden(case)
    frames, h0, weights = vc_inputs(case)
    orc = VideoChat2Oracle(weights, case.N, case.tau, case.alpha, case.sticky, case.n_layers, case.cross_freq, 12,
                           case.n_query, case.P)
    T = case.max_int
    embs = []
    for c in range(case.num_samples):
        blk = frames[c * T:(c + 1) * T].reshape(-1, case.enc_width)
        hid, emb = orc.encode_chunk(blk, h0, c == 0, chunk_uniforms(case, c))
        embs.append(emb)
        np.testing.assert_allclose(emb, g[f"c{c}_mistral"], rtol=0, atol=2e-5, err_msg=f"chunk {c}")
        if f"c{c}_hidden" in g.files:
            np.testing.assert_allclose(hid, g[f"c{c}_hidden"], rtol=0, atol=5e-5)
        for l in (0, 10):
            if f"c{c}_l{l}_along" in g.files:
                for nm in ("xq", "along", "xctx"):
                    np.testing.assert_allclose(orc.taps[f"l{l}_{nm}"], g[f"c{c}_l{l}_{nm}"], rtol=0, atol=5e-5)
        for l in range(0, case.n_layers, case.cross_freq):
            np.testing.assert_allclose(orc.ltm[l].B_past.astype(np.float64).sum(1), g[f"c{c}_l{l}_Bsum"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(np.mean(np.stack(embs), 0), g["mean_mistral"], rtol=0, atol=2e-5)
