"""Parity of the code path ``bench.py`` times: ``infv_ltm_consolidate`` at the BASELINE headline shape
(T=256, P=32, d=768, N=256, 2 layers, Q=32) with the default sub-batches (42 chunks for calls of >= 768 chunks,
28 below), the large-M projection GEMMs and the persistent chain kernel -- against the per-chunk ``forward``
chain (the per-stage kernels, which the golden tests pin to the reference) and, for its first chunks, against
the reference goldens.  Needs a real MI355X: run with ``-m gpu``.

The two paths compute the sticky probabilities with different (equally valid) fp32 association, so over
millions of draws a uniform eventually lands within rounding of a cdf edge and the free-running chains would
part there ("bit-exact index draw").  The comparison is therefore made draw by draw: the
consolidate call records every chunk's probabilities and bins (``infv_ltm_set_trace``), the per-chunk chain
resamples the SAME bins (``infv_ltm_set_bins``) while still deriving its own probabilities and its own draw;
those must agree to rounding and in all but a counted handful of draws.
"""
import os

import numpy as np
import pytest
import torch

from oracle import ltm_oracle as O
from tests.conftest import record_parity
from tests.golden.cases import CASES, call_uniforms, case_inputs, load_golden

pytestmark = pytest.mark.gpu

CTX_TOL = 1e-4
B_TOL = 2e-5
N, H, DH, D, P, T, Q, L, S = 256, 12, 64, 768, 32, 256, 32, 2, 512
DM = H * DH
HEADLINE = next(c for c in CASES if c.name == "headline")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _engine(dev, **kw):
    from infinite_video_amd.engine import LTMEngine
    return LTMEngine(N, H, DH, D, P, tau=0.75, sticky=True, n_layers=L, max_q=Q, device=dev, **kw)


def _video(dev, n_chunks):
    """Chunks 0-2 are the inputs of the ``headline`` golden case (same tokens, weights, queries, Gibbs uniforms);
    the rest is generated on the device."""
    from infinite_video_amd import synth
    ks, qs, ws = case_inputs(HEADLINE)
    projs = [tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in w) for w in ws]
    q = torch.from_numpy(np.stack(qs)).to(dev)
    k = torch.empty(n_chunks, T * P, D, device=dev, dtype=torch.float32)
    gen = torch.Generator(device=dev).manual_seed(20260)
    for i in range(0, n_chunks, 64):
        k[i:i + 64].normal_(generator=gen)
    u = synth.gibbs_uniforms(n_chunks, L, seed=20261)
    for c in range(min(3, n_chunks)):
        k[c] = torch.from_numpy(ks[c]).to(dev)
        for l in range(L):
            u[c, l] = call_uniforms(HEADLINE, c, l)
    return k, q, projs, torch.from_numpy(u).to(dev), ws, qs


def _check_against_chain(dev, fast, k, q, projs, u, ctx, bins_all, probs_all, flip_budget):
    """Per-chunk forward chain with the consolidate call's bins forced; returns (reference engine, flips)."""
    Cn = k.shape[0]
    ref = _engine(dev)
    bins_host = bins_all.cpu().numpy()
    probs_host = probs_all.cpu().numpy()
    worst = torch.zeros((), device=dev)
    flips = 0
    worst_p = 0.0
    for c in range(Cn):
        if c > 0:
            for l in range(L):
                ref.set_bins(l, bins_host[c, l])
        y = ref.forward(k[c], q, projs, u[c], new_doc=(c == 0))
        worst = torch.maximum(worst, (y - ctx[c]).abs().max())
        if c > 0:
            for l in range(L):
                own_bins, idx, own_probs = ref.last_draw(l)
                np.testing.assert_array_equal(idx, 2 * bins_host[c, l])     # left edge of bin b lies in box 2b (N = 256)
                d = own_bins != bins_host[c, l]
                flips += int(d.sum())
                assert np.abs(own_bins[d] - bins_host[c, l][d]).max(initial=0) <= 1, "a differing draw is not an adjacent bin"
                worst_p = max(worst_p, float(np.abs(own_probs / probs_host[c, l, :127] - 1).max()))
    assert float(worst) <= CTX_TOL, f"max |ctx(consolidate) - ctx(per-chunk chain)| = {float(worst):.3e}"
    assert worst_p <= 2e-5, f"sticky probabilities of the two paths differ by {worst_p:.2e} relative"
    assert flips <= flip_budget, f"{flips} of {(Cn - 1) * L * S} draws differ between the two paths"
    for l in range(L):
        np.testing.assert_allclose(fast.export_state(l)[0].cpu().numpy(), ref.export_state(l)[0].cpu().numpy(),
                                   rtol=0, atol=B_TOL)
        # the per-chunk path's last own draw against the consolidate call's diagnostics of its last step
        np.testing.assert_array_equal(fast.last_draw(l)[0], bins_host[Cn - 1, l])
        np.testing.assert_allclose(fast.last_scores(l, Q), ref.last_scores(l, Q), rtol=1e-4, atol=2e-5)
    return ref, flips


def _check_against_oracle(k, qs, ws, u, ctx, bins_all, probs_all, n_or, fast=None):
    """The CPU oracle (pinned to the reference by the goldens) over the first ``n_or`` chunks of the timed call, both
    layers, fed the call's traced bins (reference semantics: LTM.py:194-222 update, :251-286 read-out).  Per chunk:
    ctx within 1e-4, the oracle's OWN probabilities within 2e-5 relative and its own draw equal to the traced bins but
    for a counted handful of adjacent-bin flips.  With ``fast`` (the whole call was replayed): final B and last scores."""
    bins_host = bins_all[:n_or].cpu().numpy()
    probs_host = probs_all[:n_or].cpu().numpy()
    u_host = u[:n_or].cpu().numpy()
    orcs = [O.ClosedFormOracle(N, H, DH, 0.75, True, *ws[l], tokens_per_frame=P) for l in range(L)]

    def walk(l):
        """One layer's chain over the call (the layers are independent chains: one thread each, the CPU oracle's numpy / torch
        kernels release the GIL -- the 2048-chunk walk is most of the GPU suite's wall time)."""
        worst_l, worst_p_l, flips_l = 0.0, 0.0, 0
        for c in range(n_or):
            kc = k[c].cpu().numpy()
            yc = ctx[c, l].cpu().numpy()
            out = orcs[l].step(kc, qs[l], new_doc=(c == 0), u=u_host[c, l] if c else None,
                               bins_override=bins_host[c, l] if c else None)
            err = float(np.abs(out - yc).max())
            assert err <= CTX_TOL, f"chunk {c} layer {l}: |ctx(HIP) - ctx(oracle)| = {err:.3e}"
            worst_l = max(worst_l, err)
            if c:
                d = orcs[l].last_bins != bins_host[c, l]
                flips_l += int(d.sum())
                assert np.abs(orcs[l].last_bins[d] - bins_host[c, l][d]).max(initial=0) <= 1, "a differing draw is not an adjacent bin"
                worst_p_l = max(worst_p_l, float(np.abs(orcs[l].last_probs / probs_host[c, l, :127] - 1).max()))
        return worst_l, worst_p_l, flips_l

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=L) as pool:
        res = list(pool.map(walk, range(L)))                  # (an assertion inside a walk is re-raised here)
    worst = max(r[0] for r in res)
    worst_p = max(r[1] for r in res)
    flips = sum(r[2] for r in res)
    assert worst_p <= 2e-5, f"sticky probabilities of the HIP path and the oracle differ by {worst_p:.2e} relative"
    budget = max(4, int(4e-5 * n_or * L * S))
    assert flips <= budget, f"{flips} of {(n_or - 1) * L * S} oracle draws differ from the HIP path's"
    if fast is not None:
        for l in range(L):
            np.testing.assert_allclose(fast.export_state(l)[0].cpu().numpy(), orcs[l].B_past, rtol=0, atol=B_TOL)
            np.testing.assert_allclose(fast.last_scores(l, Q), orcs[l].S_prev, rtol=1e-4, atol=2e-5)
    return worst, flips


@pytest.mark.parametrize("n_chunks,x6", [(64, True), (256, True), (2048, True), (256, False), (800, False)])
def test_bench_call_matches_per_chunk_chain(dev, n_chunks, x6, monkeypatch):
    """64 chunks: two sub-batches + a tail (large-M GEMM branch, persistent chain); 256 chunks: one 8-GPU
    shard; 2048 chunks with max_batch_chunks=42: exactly bench.py's call.  x6 (the default): the projection GEMM as six bf16 MFMA
    products of exact three-piece splits; not x6 (INFV_PROJ_X6=0): the fp32-MFMA GEMM of rounds 1-3 -- same goldens, same oracle,
    same budgets (round 6: its long call is 800 chunks -- nineteen 42-chunk sub-batches + a short one, past the wrap of the five
    workspace sets, long enough for the one pooling launch per call -- instead of 2048: the GPU suite's wall time).  The observed flip counts are recorded (conftest.record_parity)."""
    k, q, projs, u, ws, qs = _video(dev, n_chunks)
    if not x6:
        monkeypatch.setenv("INFV_PROJ_X6", "0")
    fast = _engine(dev, max_batch_chunks=42)
    monkeypatch.delenv("INFV_PROJ_X6", raising=False)
    bins_all, probs_all = fast.set_trace(n_chunks)
    ctx = fast.consolidate(k, q, projs, u, new_doc=True)
    fast.sync()
    fast.set_trace(0)
    assert bool(torch.isfinite(ctx).all())
    assert int((bins_all[1:] < 0).sum()) == 0 and int((bins_all[0] >= 0).sum()) == 0   # every chunk but the first drew
    # reproducible run to run (fixed-point histogram: no order dependence)
    again = fast.consolidate(k, q, projs, u, new_doc=True)
    assert torch.equal(again, ctx)
    # chunks 0-2 are the reference's own run of the headline golden case
    g = load_golden(HEADLINE)
    for c in range(3):
        for l in range(L):
            np.testing.assert_allclose(ctx[c, l].cpu().numpy(), g[f"c{c}_l{l}_ctx"], rtol=0, atol=CTX_TOL)
            if c > 0:
                np.testing.assert_array_equal(bins_all[c, l].cpu().numpy(), g[f"c{c}_l{l}_bins"])
                np.testing.assert_allclose(probs_all[c, l, :127].cpu().numpy(), g[f"c{c}_l{l}_probs"], rtol=2e-5, atol=1e-9)
    budget = max(4, int(4e-5 * n_chunks * L * S))
    ref, flips = _check_against_chain(dev, fast, k, q, projs, u, ctx, bins_all, probs_all, budget)
    record_parity(f"[timed path] x6={x6} {n_chunks} chunks: {flips} of {(n_chunks - 1) * L * S} draws differ between consolidate and the per-chunk chain")
    # the CPU oracle itself, both layers, teacher-forced to the call's traced bins, over the WHOLE call (then also final B
    # and last scores) -- including the 2048-chunk bench call (ring wrap of the workspace / R sets, the short 49th
    # sub-batch with its own projection dispatch: all past chunk 128; about a minute of CPU) on the path bench.py times,
    # the DEFAULT (bf16x6) projection GEMM; the legacy fp32-MFMA variant (INFV_PROJ_X6=0) of the 2048-chunk call keeps
    # the first 128 chunks (three 42-chunk sub-batches)
    n_or = n_chunks if (n_chunks <= 800 or x6) else 128
    worst, oflips = _check_against_oracle(k, qs, ws, u, ctx, bins_all, probs_all, n_or, fast if n_or == n_chunks else None)
    record_parity(f"[timed path] x6={x6} {n_chunks} chunks vs the CPU oracle over {n_or}: max |ctx diff| {worst:.2e}, {oflips} draws differ")


def test_split_bf16_value_projection_stays_inside_tolerance(dev, monkeypatch):
    """INFV_VPROJ_SPLIT=1 (off by default): the V' half of a sub-batch's projection as three bf16 MFMA products.
    Same draws (the score half stays exact fp32), read-out within the 1e-4 gate of the exact-fp32 run."""
    k, q, projs, u, _, _ = _video(dev, 64)
    exact = _engine(dev, max_batch_chunks=42)
    be, _ = exact.set_trace(64)
    ctx_e = exact.consolidate(k, q, projs, u, new_doc=True)
    monkeypatch.setenv("INFV_VPROJ_SPLIT", "1")
    split = _engine(dev, max_batch_chunks=42)
    monkeypatch.delenv("INFV_VPROJ_SPLIT")
    bs, _ = split.set_trace(64)
    ctx_s = split.consolidate(k, q, projs, u, new_doc=True)
    split.sync()
    assert torch.equal(be, bs)
    err = float((ctx_e - ctx_s).abs().max())
    assert 0.0 < err <= CTX_TOL, err            # > 0: the split path really ran
    for l in range(L):
        np.testing.assert_array_equal(exact.export_state(l)[0].cpu().numpy(), split.export_state(l)[0].cpu().numpy())


_FAULT_CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from infinite_video_amd import _lib
from tests.test_timed_path_gpu import _engine, _video
dev = torch.device("cuda:0")
k, q, projs, u, _, _ = _video(dev, 8)
os.environ["INFV_CHAIN_FAULT"] = "1"                          # read when a handle is created (experiments build only)
bad = _engine(dev, max_batch_chunks=4)
del os.environ["INFV_CHAIN_FAULT"]
bad.consolidate(k, q, projs, u, new_doc=True)
try:
    bad.sync()
    raise SystemExit("the time-out was swallowed")
except _lib.LTMError as e:
    assert e.code == -4 and "timed out" in str(e), str(e)
assert not bad.has_memory                                   # the invalid memory was dropped
bad.sync()                                                  # reported once
bad.consolidate(k, q, projs, u, new_doc=True)
torch.cuda.synchronize()
try:
    bad.export_state(0)                                     # ... and caught by whatever entry point comes next
    raise SystemExit("the second time-out was swallowed")
except _lib.LTMError:
    pass
good = _engine(dev, max_batch_chunks=4)
ctx = good.consolidate(k, q, projs, u, new_doc=True)
good.sync()
assert bool(torch.isfinite(ctx).all())
print("FAULT_CHILD_OK")
'''


def _run_child(code, env_add, *args, timeout=900):
    import subprocess
    import sys
    env = dict(os.environ, **env_add)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, "-c", code, *args], check=True, env=env, cwd=root, timeout=timeout,
                          capture_output=True, text=True)


def test_chain_timeout_is_reported_not_swallowed(dev):
    """Fault injection (experiments build of the library, INFV_LTM_LIBRARY=exp: the shipped one has no such switch): the
    persistent chain kernel is told to expect one arrival more than its workgroups can deliver, so every wait times out.
    The failure must surface as INFV_ERR_STATE at sync() (and at the next entry point), once, and a fresh engine must be
    unaffected.  Runs in a child process: a process loads one build of the library."""
    r = _run_child(_FAULT_CHILD, {"INFV_LTM_LIBRARY": "exp"})
    assert "FAULT_CHILD_OK" in r.stdout, r.stdout + r.stderr
    # the same with round 5's call-long launches (role S, pooling and projection GEMM resident for the whole call, handed off through
    # device-side counters): every bounded wait -- exchange, readiness poll, flag_wait_kernel -- must give up, latch and report
    r = _run_child(_FAULT_CHILD, {"INFV_LTM_LIBRARY": "exp", "INFV_CHAIN_CALL": "1", "INFV_POOL_CALL": "1", "INFV_GEMM_CALL": "1"})
    assert "FAULT_CHILD_OK" in r.stdout, r.stdout + r.stderr


def test_consolidate_video_through_rccl_world_of_one(dev):
    """The multi-GPU entry point with an initialised "nccl" (= RCCL) group of one rank: the all-gather branch runs on
    the device and returns this rank's own memory."""
    import socket
    import torch.distributed as dist
    from infinite_video_amd.video_memory import consolidate_video
    k, q, projs, u, _, _ = _video(dev, 6)
    eng = _engine(dev, max_batch_chunks=4)
    ctx0, mem0 = consolidate_video(eng, k, q, projs, u)         # no group: no collective
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=dev)
    try:
        ctx1, mem1 = consolidate_video(eng, k, q, projs, u)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    assert torch.equal(ctx0, ctx1)
    assert mem1.B.shape == (1, L, N, D) and float(mem1.count[0]) == 6.0
    assert torch.equal(mem0.B, mem1.B) and torch.equal(mem0.bin_mass, mem1.bin_mass)
    torch.testing.assert_close(mem1.mean_embedding(), ctx1.mean(0), rtol=1e-5, atol=1e-6)


_VARIANT_CHILD = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from tests.test_timed_path_gpu import _engine, _video
dev = torch.device("cuda:0")
k, q, projs, u, _, _ = _video(dev, 75)
e = _engine(dev, max_batch_chunks=42)
a = e.consolidate(k[:70], q, projs, u[:70], new_doc=True).clone()
b = e.consolidate(k[70:], q, projs, u[70:], new_doc=False).clone()
e.sync()
B = [e.export_state(l)[0].cpu().numpy() for l in range(2)]
np.savez(sys.argv[1], a=a.cpu().numpy(), b=b.cpu().numpy(), B0=B[0], B1=B[1], bins=e.last_draw(0)[0])
'''

# non-default variants kept in the sources (A/B material, experiments build only): each must reproduce the shipped default
_CALL = {"INFV_CHAIN_CALL": "1"}                                # role S as ONE launch per call (round 5), atomics exchange
_VARIANTS = [
    {},                                                        # the experiments build with no knob set
    {"INFV_POOL_ROWS": "0"},                                   # pool_frames_kernel + build_rows_kernel (the round-2 form)
    {"INFV_POOL_ROWS": "2", "INFV_PR_U": "4", "INFV_PR_WGS": "500"},   # default kernel, 4-load bursts, grid-stride (rows only: split3_rows_kernel makes the planes)
    {"INFV_POOL_PLANES": "0"},                                 # the pooling kernel writes rows only, split3_rows_kernel makes the GEMM's bf16 planes (round 4; what calls under 768 chunks ship since round 6)
    {"INFV_POOL_PLANES": "2"},                                 # the pooling kernel writes the planes too (what calls of 768+ chunks ship), here on a short call
    _CALL,                                                     # call-long role S; pooling and GEMM launched per sub-batch
    dict(_CALL, INFV_POOL_CALL="1"),                           # ... + ONE pooling launch per call (completion counts instead of launch boundaries)
    dict(_CALL, INFV_POOL_CALL="1", INFV_GEMM_CALL="1"),       # ... + the projection GEMM as a resident tile-queue kernel (32 workgroups)
    dict(_CALL, INFV_POOL_CALL="1", INFV_GEMM_CALL="1", INFV_GEMM_WGS="7"),   # ... with 7 workgroups: a sub-batch's tiles over several rounds
    {"INFV_CHAIN_RPW": "1"},                                   # 8-row chain tiles (96 workgroups) as in round 2
    {"INFV_PROJ_X6": "0", "INFV_VPROJ_ON_UC": "1"},            # (fp32-MFMA GEMM) V' half of the projection as its own GEMM on the UC stream
    dict(_CALL, INFV_PROJ_X6="0", INFV_POOL_CALL="1"),         # (fp32-MFMA GEMM) call-long role S and pooling launch (rows only)
    {"INFV_PERSISTENT": "0"},                                  # one role-S launch per chunk
    {"INFV_PROJ_X6": "0", "INFV_GEMM_LW": "0"},                # (fp32-MFMA GEMM) without loader waves
    {"INFV_POOL_TID": "1"},                                    # pooling kernel with lane-id addressed loads (no vector address operand)
    {"INFV_PROJ_X6": "0", "INFV_GEMM_SLICES": "2"},            # (fp32-MFMA GEMM) launched as two column slices
    {"INFV_POOL_PRIO": "1", "INFV_UC_PRIO": "2", "INFV_ALPHA_PRIO": "2", "INFV_WG_STAMPS": "1"},   # wave priorities + residency stamps
    {"INFV_ALPHA_DIRECT": "1", "INFV_ALPHA_UPW": "5"},          # alpha_rows2_kernel's fallback staging (shapes beyond its register stage), 5 units per workgroup
    {"INFV_POOL_CALL": "2"},                                   # ONE pooling launch for the call beside per-sub-batch role-S / GEMM launches: what calls of 768+ chunks ship since round 6, here forced on a short one
    {"INFV_POOL_CALL": "2", "INFV_DROP_WAITS": "0"},           # ... with the (redundant) wait of the caller's stream for the UC kernel of five sub-batches ago, as up to round 5
    {"INFV_SMALL_TILES": "5"},                                 # the projection of the first and the last sub-batch as 128 x 128 tiles (what the last sub-batch of a 768+-chunk call ships, round 6)
    {"INFV_CHAIN_DMA": "1"},                                   # role S with the LDS-DMA loader (128 registers, round 6); the short last sub-batch keeps the register loader
]


def test_kept_variants_reproduce_the_default_bit_for_bit(dev, tmp_path):
    """A 70-chunk call (two 32-chunk sub-batches + a short one: split-K slabs) and a 5-chunk continuation, once with the
    shipped library and once per variant with the experiments build: contexts, memory and draws identical bit for bit (variants of
    the fp32-MFMA projection GEMM against the shipped library with INFV_PROJ_X6=0).  Includes round 5's call-long launches of
    role S, of the pooling and of the projection GEMM: same arithmetic, other hand-offs (device-side flags instead of launch
    boundaries)."""
    def run(env_add, name):
        path = str(tmp_path / name)
        _run_child(_VARIANT_CHILD, env_add, path)
        return {k_: v for k_, v in np.load(path).items()}
    base = {"": run({}, "base.npz"), "0": run({"INFV_PROJ_X6": "0"}, "base_f32mfma.npz")}   # shipped library: default, and the fp32-MFMA GEMM
    for i, v in enumerate(_VARIANTS):
        got = run(dict(v, INFV_LTM_LIBRARY="exp"), f"v{i}.npz")
        want = base[v.get("INFV_PROJ_X6", "")]
        for key in want:
            if "INFV_PERSISTENT" in v and key in ("a", "b"):
                # one role-S launch per chunk computes the read-out weights in the chain kernel itself (a lane per box n, n + 64, ..);
                # alpha_rows2_kernel gives a lane the boxes 4 lane .. 4 lane + 3: another order of the same 256-term fp32 sum
                np.testing.assert_allclose(want[key], got[key], rtol=0, atol=2e-7, err_msg=f"{v}: {key}")
            else:
                np.testing.assert_array_equal(want[key], got[key], err_msg=f"{v}: {key}")


def test_mailbox_exchange_variants(dev, tmp_path):
    """INFV_CHAIN_XCD=1 (experiments build): role S exchanges its bin masses through mailboxes (inside one XCD's L2 with the
    XCD-aware grid + placement handshake, csrc/ltm_chain_batch.hip).  The totals are fp32 sums of per-workgroup row sums in a fixed
    order instead of the exact fixed-point totals of the atomics exchange: same draws (but for a flip within rounding of a cdf
    edge), contexts and memory equal to fp32 rounding; the result must not depend on the placement (sc1 mailboxes when the
    handshake finds the layer on several XCDs: forced with INFV_S_FLAGS=32, or a linear grid: INFV_CHAIN_LINEAR=1) nor on whether
    role S is launched per sub-batch or once per call -- bit for bit."""
    def run(env_add, name):
        path = str(tmp_path / name)
        _run_child(_VARIANT_CHILD, env_add, path)
        return {k_: v for k_, v in np.load(path).items()}
    base = run({}, "base.npz")
    xcd = run({"INFV_LTM_LIBRARY": "exp", "INFV_CHAIN_XCD": "1"}, "xcd.npz")
    for name, env in (("far", {"INFV_S_FLAGS": "32"}), ("linear", {"INFV_CHAIN_LINEAR": "1"}), ("call", {"INFV_CHAIN_CALL": "1"}),
                      ("call_linear", {"INFV_CHAIN_CALL": "1", "INFV_CHAIN_LINEAR": "1", "INFV_POOL_CALL": "1", "INFV_GEMM_CALL": "1"})):
        other = run(dict(env, INFV_LTM_LIBRARY="exp", INFV_CHAIN_XCD="1"), name + ".npz")
        for key in base:
            np.testing.assert_array_equal(xcd[key], other[key], err_msg=f"{name} changed {key}")
    assert int((base["bins"] != xcd["bins"]).sum()) <= 1
    for key in ("a", "b"):
        assert float(np.abs(base[key] - xcd[key]).max()) <= 1e-5, key
    for key in ("B0", "B1"):
        assert float(np.abs(base[key] - xcd[key]).max()) <= 2e-6, key


@pytest.mark.parametrize("n_chunks,max_batch,split", [(33, 42, 0), (45, 7, 0), (70, 28, 37), (129, 42, 1), (97, 13, 50),
                                                       (300, 42, 0), (64, 32, 63), (800, 42, 0), (790, 30, 20)])
def test_odd_call_lengths_sub_batches_and_continuations(dev, n_chunks, max_batch, split):
    """Call lengths that are no multiple of the sub-batch, short last sub-batches (split-K slabs), one-chunk calls and a
    document continued by a second consolidate call (``new_doc=False``): every one must equal the per-chunk chain.  800 / 790
    chunks: calls long enough for the one pooling launch per call (768+ chunks, round 6) with a short last sub-batch / as the
    continuation of a 20-chunk call."""
    k, q, projs, u, ws, qs = _video(dev, n_chunks)
    fast = _engine(dev, max_batch_chunks=max_batch)
    pieces = [(0, n_chunks)] if split == 0 else [(0, split), (split, n_chunks)]
    ctxs, bins, probs = [], [], []
    for lo, hi in pieces:
        b, p = fast.set_trace(hi - lo)
        ctxs.append(fast.consolidate(k[lo:hi], q, projs, u[lo:hi], new_doc=(lo == 0)).clone())
        fast.sync()
        bins.append(b.clone()); probs.append(p.clone())
    fast.set_trace(0)
    ctx, bins_all, probs_all = torch.cat(ctxs), torch.cat(bins), torch.cat(probs)
    assert int((bins_all[1:] < 0).sum()) == 0 and int((bins_all[0] >= 0).sum()) == 0
    budget = max(4, int(4e-5 * n_chunks * L * S))
    _check_against_chain(dev, fast, k, q, projs, u, ctx, bins_all, probs_all, budget)


@pytest.mark.parametrize("n_query", [16, 24, 5])
def test_fewer_query_rows_than_a_full_tile(dev, n_query):
    """Q < 32: the fast path (Q = 16: half-empty read-out tiles in the UC kernel, one 8-row chain tile fewer) and the
    shapes it hands to the per-chunk path (Q = 24, 5) must equal the per-chunk forward chain."""
    k, q, projs, u, _, _ = _video(dev, 40)
    qq = q[:, :n_query].contiguous()
    a, b = _engine(dev, max_batch_chunks=42), _engine(dev, max_batch_chunks=42)
    fast = a.consolidate(k, qq, projs, u, new_doc=True)
    a.sync()
    ref = torch.stack([b.forward(k[c], qq, projs, u[c], new_doc=(c == 0)) for c in range(k.shape[0])])
    assert float((fast - ref).abs().max()) <= CTX_TOL
    for l in range(L):
        np.testing.assert_array_equal(a.last_draw(l)[0], b.last_draw(l)[0])
        np.testing.assert_allclose(a.export_state(l)[0].cpu().numpy(), b.export_state(l)[0].cpu().numpy(), rtol=0, atol=B_TOL)
