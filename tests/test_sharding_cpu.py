"""Multi-GPU sharding logic on CPU: world_size 2 over gloo, with an oracle-backed stand-in for the
HIP engine (the product engine has no CPU path; the collective and the bookkeeping are what is
under test here)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from infinite_video_amd import synth
from infinite_video_amd.video_memory import (consolidate_video, pack_local_memory, shard_range, unpack_memory)
from oracle.ltm_oracle import ClosedFormOracle

N, H, DH, D, P, T, Q, L, C = 64, 12, 64, 768, 32, 8, 32, 2, 6
DM = H * DH


class OracleEngine:
    """Same surface as infinite_video_amd.engine.LTMEngine (consolidate / export_state), CPU oracle inside."""

    def __init__(self):
        self.L, self.N, self.d, self.dm, self.H = L, N, D, DM, H
        self.ws = [synth.layer_projections(l, D, DM) for l in range(L)]
        self.layers = [ClosedFormOracle(N, H, DH, .75, True, *self.ws[l], tokens_per_frame=P) for l in range(L)]

    def consolidate(self, k, q, projs, u, new_doc=True):
        out = np.zeros((k.shape[0], L, Q, DM), np.float32)
        for c in range(k.shape[0]):
            for l in range(L):
                out[c, l] = self.layers[l].step(k[c].numpy(), q[l].numpy(), new_doc=(new_doc and c == 0), u=u[c, l].numpy())
        return torch.from_numpy(out)

    def sync(self):
        pass

    # chain state of the stand-in: B_past and the last scores of every layer (the oracle derives everything else)
    def chain_state_numel(self, Q):
        return L * (N * D + H * Q * N)

    def export_chain_state(self, Q):
        return torch.cat([torch.from_numpy(np.concatenate([o.B_past.reshape(-1), o.S_prev.reshape(-1)])) for o in self.layers])

    def import_chain_state(self, Q, blob):
        per = N * D + H * Q * N
        for l, o in enumerate(self.layers):
            part = blob[l * per:(l + 1) * per].numpy()
            o.B_past = part[:N * D].reshape(N, D).copy()
            o.S_prev = part[N * D:].reshape(H, Q, N).copy()

    def last_scores_device(self, Q):
        return torch.from_numpy(np.stack([o.S_prev for o in self.layers]).astype(np.float32))

    def export_state(self, l):
        o = self.layers[l]
        return torch.from_numpy(o.B_past), torch.from_numpy(o.sticky_p_raw(o.S_prev).astype(np.float32))


def _inputs():
    k = torch.from_numpy(np.stack([synth.frame_tokens(c, T, P, D) for c in range(C)]))
    q = torch.from_numpy(np.stack([synth.layer_query(l, Q, DM) for l in range(L)]))
    u = torch.from_numpy(synth.gibbs_uniforms(C, L))
    return k, q, u


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 2048):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def test_pack_unpack_roundtrip_single_process():
    eng = OracleEngine()
    k, q, u = _inputs()
    ctx, mem = consolidate_video(eng, k[:3], q, None, u[:3])
    assert mem.B.shape == (1, L, N, D) and mem.bin_mass.shape == (1, L, 127) and mem.ctx_sum.shape == (1, L, Q, DM)
    assert float(mem.count[0]) == 3.0
    torch.testing.assert_close(mem.mean_embedding(), ctx.mean(0))
    torch.testing.assert_close(mem.B[0, 1], eng.export_state(1)[0])
    with pytest.raises(ValueError):
        unpack_memory(pack_local_memory(eng, ctx)[:-1], 1, L, N, D, Q, DM)
    assert mem.scores is None
    # the full payload of SURVEY.md section 8e: + the last scores [L, H, Q, N]
    ctx2, mem2 = consolidate_video(OracleEngine(), k[:3], q, None, u[:3], with_scores=True)
    assert mem2.scores.shape == (1, L, H, Q, N)
    torch.testing.assert_close(mem2.scores[0, 1], torch.from_numpy(eng.layers[1].S_prev.astype(np.float32)))
    torch.testing.assert_close(mem2.B, mem.B)
    torch.testing.assert_close(mem2.ctx_sum, mem.ctx_sum)
    with pytest.raises(ValueError):
        unpack_memory(pack_local_memory(eng, ctx, True), 1, L, N, D, Q, DM)          # scores in the payload, not announced


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        k, q, u = _inputs()
        a, b = shard_range(C, world, rank)
        ctx, mem = consolidate_video(OracleEngine(), k[a:b], q, None, u[a:b])
        ret[rank] = (ctx.numpy(), mem.B.numpy(), mem.bin_mass.numpy(), mem.ctx_sum.numpy(), mem.count.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_equals_single_process_subvideos():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    k, q, u = _inputs()
    for rank in range(2):
        a, b = shard_range(C, 2, rank)
        ref = OracleEngine()
        ctx_ref = ref.consolidate(k[a:b], q, None, u[a:b])        # rank's block == its own document
        ctx, B, mass, ctx_sum, count = ret[rank]
        np.testing.assert_array_equal(ctx, ctx_ref.numpy())
        # every rank holds every rank's memory after the all-gather
        for other in range(2):
            B0, m0, s0, c0 = ret[other][1], ret[other][2], ret[other][3], ret[other][4]
            np.testing.assert_array_equal(B, B0); np.testing.assert_array_equal(mass, m0)
            np.testing.assert_array_equal(ctx_sum, s0); np.testing.assert_array_equal(count, c0)
        np.testing.assert_array_equal(B[rank, 0], ref.export_state(0)[0].numpy())
        np.testing.assert_allclose(ctx_sum[rank], ctx_ref.sum(0).numpy(), rtol=1e-6, atol=1e-6)
    assert list(ret[0][4]) == [3.0, 3.0]


def _handoff_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        k, q, u = _inputs()
        a, b = shard_range(C, world, rank)
        ctx, mem = consolidate_video(OracleEngine(), k[a:b], q, None, u[a:b], handoff=True)
        ret[rank] = (ctx.numpy(), mem.B.numpy(), mem.count.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_handoff_equals_the_single_stream_run_bit_for_bit():
    """consolidate_video(handoff=True): rank 1 continues rank 0's memory chain (send/recv of the chain state over gloo);
    the concatenated outputs and the final memory equal ONE process walking the whole video -- the reference's loop."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_handoff_worker, args=(2, port, ret), nprocs=2, join=True)
    k, q, u = _inputs()
    single = OracleEngine()
    ctx_ref = single.consolidate(k, q, None, u).numpy()
    got = np.concatenate([ret[0][0], ret[1][0]])
    np.testing.assert_array_equal(got, ctx_ref)
    # the last rank's memory is the single-stream memory; rank 0's is the memory at the block boundary
    np.testing.assert_array_equal(ret[0][1][1, 0], single.export_state(0)[0].numpy())
    assert list(ret[1][2]) == [3.0, 3.0]
    # and it differs from the independent-documents mode (the default), whose rank 1 starts from an empty memory
    indep = OracleEngine()
    a, b = shard_range(C, 2, 1)
    assert np.abs(indep.consolidate(k[a:b], q, None, u[a:b]).numpy() - ctx_ref[a:b]).max() > 1e-3


def _subgroup_handoff_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        grp = dist.new_group(ranks=[1, 2])               # every process creates it; global rank 0 is not a member
        if rank == 0:
            return
        k, q, u = _inputs()
        a, b = shard_range(C, 2, dist.get_rank(grp))
        ctx, mem = consolidate_video(OracleEngine(), k[a:b], q, None, u[a:b], group=grp, handoff=True)
        ret[rank] = (ctx.numpy(), mem.B.numpy(), mem.count.numpy())
    finally:
        dist.destroy_process_group()


def test_handoff_inside_a_subgroup_whose_ranks_are_not_the_global_ones():
    """The chain state travels between GROUP ranks 0 -> 1, which are global ranks 1 -> 2 here: send / recv take
    global ranks, so the hand-off must translate (it went to the wrong process, or hung, before)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_subgroup_handoff_worker, args=(3, port, ret), nprocs=3, join=True)
    k, q, u = _inputs()
    ctx_ref = OracleEngine().consolidate(k, q, None, u).numpy()
    np.testing.assert_array_equal(np.concatenate([ret[1][0], ret[2][0]]), ctx_ref)
    assert list(ret[2][2]) == [3.0, 3.0]


def _qf_gather_worker(rank, world, port, out_q):
    import torch
    import torch.distributed as dist
    from infinite_video_amd.video_qformer import gather_video_embeddings
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(7)
    embs = torch.randn(5, 1, 4, 8, generator=g)                  # 5 chunks of the "video": rank 0 takes 3, rank 1 takes 2
    mine = embs[:3] if rank == 0 else embs[3:]
    mem = [torch.full((2, 3), float(rank)), torch.full((4,), 10.0 + rank)]
    mean, per_rank, counts = gather_video_embeddings(mine.sum(0), mine.size(0), mem)
    ok = torch.allclose(mean, embs.mean(0), atol=1e-6) and counts.tolist() == [3.0, 2.0]
    ok = ok and all(float(per_rank[r][0][0, 0]) == float(r) and float(per_rank[r][1][0]) == 10.0 + r for r in range(world))
    ok = ok and tuple(per_rank[1][0].shape) == (2, 3)
    out_q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_qformer_block_sums_reproduce_the_global_mean_over_chunks():
    """world_size 2, gloo: uneven chunk blocks; sum + count per rank == the eval loop's plain mean over chunks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_qf_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_bench_rank_body_end_to_end_with_a_stub_engine():
    """``python bench.py --gpus 2 --stub-engine``: the parent spawns two ranks through torch.distributed.run (127.0.0.1
    rendezvous), each shards the video, runs warm-up + timed steps through consolidate_video (gloo all-gather), the clock is the
    max over ranks and EXACTLY ONE JSON line appears on stdout, printed by rank 0, carrying the shard / all-gather decomposition
    a SCALE line needs.  The engine is a stand-in (no GPU here); everything around it is the code the 8-GPU run executes."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stub-engine", "--chunks", "6",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["stub_engine"] is True
    assert d["config"]["chunks"] == 6 and d["config"]["chunks_per_gpu"] == 3
    assert d["gathered_counts"] == [3.0, 3.0]                      # every rank's block arrived in the all-gather
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["shard_ms"] > 0 and d["allgather_ms"] >= 0
