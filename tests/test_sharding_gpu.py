"""Two processes on ONE GPU (fresh children, backend gloo, a real LTMEngine in each): engine + sharding + collective
together, which the CPU test (oracle stand-in) and the world-of-one RCCL test never combine.

  * default mode: rank r's outputs equal a single-process run on rank r's sub-video bit for bit;
  * ``handoff=True``: the concatenated outputs and the final memory equal ONE process walking the whole video bit for bit
    (the reference's loop, long_term_attention_gibbs.py:194-222) -- the correctness mode.
Needs a real MI355X: run with ``-m gpu``."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# 129 chunks -> blocks of 65 and 64: every sub-batch of the single-stream run and of both ranks is a full 32-chunk one
# (no split-K slabs for a short last sub-batch, whose different summation order would show in the last bit)
N_CHUNKS = 129


def _run_world(mode, tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, "-m", "tests.shard_worker", str(r), "2", str(port), mode, str(N_CHUNKS), str(tmp_path)],
                              cwd=ROOT, env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=900) == 0, f"rank process failed ({mode})"
    return [dict(np.load(os.path.join(tmp_path, f"rank{r}.npz"))) for r in range(2)]


@pytest.fixture(scope="module")
def single():
    """Single-process runs in the pytest process: the whole video, and each rank's block as its own document."""
    from infinite_video_amd.video_memory import shard_range
    from tests.test_timed_path_gpu import L, _engine, _video
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    k, q, projs, u, _, _ = _video(dev, N_CHUNKS)
    out = {}
    eng = _engine(dev, max_batch_chunks=42)
    out["whole"] = eng.consolidate(k, q, projs, u, new_doc=True).cpu().numpy()
    eng.sync()
    out["whole_B"] = np.stack([eng.export_state(l)[0].cpu().numpy() for l in range(L)])
    out["whole_bins"] = np.stack([eng.last_draw(l)[0] for l in range(L)])
    for r in range(2):
        a, b = shard_range(N_CHUNKS, 2, r)
        e = _engine(dev, max_batch_chunks=42)
        out[f"block{r}"] = e.consolidate(k[a:b].contiguous(), q, projs, u[a:b].contiguous(), new_doc=True).cpu().numpy()
        e.sync()
        out[f"block{r}_B"] = np.stack([e.export_state(l)[0].cpu().numpy() for l in range(L)])
    del k
    torch.cuda.empty_cache()
    return out


def test_two_processes_one_gpu_equal_single_process_subvideos(single, tmp_path):
    ranks = _run_world("shard", tmp_path)
    for r in range(2):
        np.testing.assert_array_equal(ranks[r]["ctx"], single[f"block{r}"])
        # after the all-gather every rank holds every rank's memory
        for o in range(2):
            np.testing.assert_array_equal(ranks[r]["B"][o], single[f"block{o}_B"])
        np.testing.assert_array_equal(ranks[r]["count"], np.array([65.0, 64.0], np.float32))
        np.testing.assert_allclose(ranks[r]["ctx_sum"][r], single[f"block{r}"].sum(0), rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(ranks[0]["B"], ranks[1]["B"])
    # sharding changes the result near the block boundary (the default mode's documented caveat)
    assert np.abs(ranks[1]["ctx"][0] - single["whole"][65]).max() > 1e-3


def test_handoff_mode_reproduces_the_single_stream_run_bit_for_bit(single, tmp_path):
    ranks = _run_world("handoff", tmp_path)
    got = np.concatenate([ranks[0]["ctx"], ranks[1]["ctx"]])
    np.testing.assert_array_equal(got, single["whole"])
    np.testing.assert_array_equal(ranks[1]["B"][1], single["whole_B"])      # the last rank ends with the single-stream memory
    np.testing.assert_array_equal(ranks[1]["bins"], single["whole_bins"])


def test_chain_state_blob_of_another_shape_is_refused():
    """The blob of infv_ltm_export_chain_state starts with a header (magic, version, L, N, d, dm, H, Q, n_bins).  import checks it on
    the device, so the call stays asynchronous: a blob exported by a handle of another shape latches an error that the next
    entry point returns (INFV_ERR_STATE), and the importing handle's memory is reset; a matching blob continues the chain."""
    from infinite_video_amd import _lib
    from infinite_video_amd.engine import LTMEngine
    from tests.test_timed_path_gpu import L, _engine, _video
    dev = torch.device("cuda:0")
    k, q, projs, u, _, _ = _video(dev, 64)                                # (blocks of 32 chunks = the call's sub-batches: see consolidate_video)
    a = _engine(dev, max_batch_chunks=42)
    a.consolidate(k[:32], q, projs, u[:32], new_doc=True)
    a.sync()
    Q = q.shape[1]
    blob = a.export_chain_state(Q)
    hdr = blob[:16].view(torch.int32).cpu().numpy()
    assert hdr[0] == 0x43464E49 and hdr[1] == 1 and list(hdr[2:4]) == [L, 256] and hdr[7] == Q
    ok = _engine(dev, max_batch_chunks=42)
    ok.import_chain_state(Q, blob)
    cont = ok.consolidate(k[32:], q, projs, u[32:], new_doc=False)
    whole = _engine(dev, max_batch_chunks=42).consolidate(k, q, projs, u, new_doc=True)    # one call over all 64 chunks
    assert torch.equal(cont, whole[32:])
    bad = blob.clone()
    bad[:16].view(torch.int32)[3] = 128                                  # "exported by a handle with N = 128"
    other = _engine(dev, max_batch_chunks=42)
    other.import_chain_state(Q, bad)
    torch.cuda.synchronize()
    with pytest.raises(_lib.LTMError) as ei:
        other.consolidate(k[32:], q, projs, u[32:], new_doc=False)
    assert ei.value.code == -4 or "blob" in str(ei.value)                # INFV_ERR_STATE
    assert not other.has_memory


def test_gathered_memory_can_carry_the_last_scores():
    """``consolidate_video(..., with_scores=True)``: SURVEY.md section 8e's full payload -- B_past, the last scores [L, H, Q, N], the sum
    of the outputs and the count.  The scores are the ones the handle's diagnostics report (``last_scores``) without their bias
    term, the rest of the gathered memory is unchanged."""
    from infinite_video_amd.video_memory import consolidate_video
    from tests.test_timed_path_gpu import H, L, N, Q, _engine, _video
    dev = torch.device("cuda:0")
    k, q, projs, u, _, _ = _video(dev, 40)
    eng = _engine(dev, max_batch_chunks=42)
    ctx0, mem0 = consolidate_video(eng, k, q, projs, u)
    ctx1, mem1 = consolidate_video(eng, k, q, projs, u, with_scores=True)
    torch.cuda.synchronize()
    assert mem0.scores is None and mem1.scores.shape == (1, L, H, Q, N)
    assert torch.equal(ctx0, ctx1) and torch.equal(mem0.B, mem1.B) and torch.equal(mem0.bin_mass, mem1.bin_mass)
    assert torch.equal(mem0.ctx_sum, mem1.ctx_sum)
    # the blob's scores are bias-free (q . K'^T / sqrt(dh)); the diagnostics add the bias term q_h . bk_h / sqrt(dh), constant along N
    from tests.test_timed_path_gpu import DH
    for l in range(L):
        got = mem1.scores[0, l].cpu().numpy().astype(np.float64)
        full = np.asarray(eng.last_scores(l, Q), dtype=np.float64)
        bk = projs[l][1].cpu().numpy().astype(np.float64).reshape(H, DH)
        cq = (q[l].cpu().numpy().astype(np.float64).reshape(Q, H, DH) * bk[None]).sum(-1).T / np.sqrt(DH)      # [H, Q]
        np.testing.assert_allclose(got + cq[:, :, None], full, rtol=0, atol=2e-6)
