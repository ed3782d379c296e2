#!/usr/bin/env python3
"""Headline benchmark: frame-chunks/sec consolidated (max_int=256, num_basis=256, d=768).

One "step" = one pass of the LTM consolidation path over a synthetic video of ``--chunks``
chunks (default 2048, BASELINE.json configs[1]/[2]): every chunk's frame tokens [256*32, 768]
are mean-pooled, regressed onto 256 box bases, the sticky (Gibbs-sampled) memory of BOTH video
Q-former LTM layers is updated and read out.  The Q-former/LLM are stubbed (the op is called
directly; with alpha=1.0 the reference bypasses it, Qformer.py:220-223).  Inputs are resident
in HBM before the timed region.  With N > 1 ranks the video is cut into contiguous blocks
(strong scaling: total work fixed), each rank consolidates its block as its own document and one
RCCL all-gather exchanges the consolidated memories (infinite_video_amd.video_memory).

Launch: ``python bench.py --gpus N``.  With N > 1 and no WORLD_SIZE in the environment this
process only spawns ``python -m torch.distributed.run --nproc-per-node N ... bench.py`` (it never
touches a GPU itself) and forwards the child's output; under torchrun it is one rank.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field meanings).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md), the graded roofline
SELFCHECK_MAX_ABS_ERR = 1e-3   # north-star tolerance of the contexts (fp32)
SELFCHECK_MAX_FLIPS = 8        # draws (of 25 chunks x 2 layers x 512) allowed to land in the adjacent bin
T, P, D, N, H, DH, Q, L, TAU, S = 256, 32, 768, 256, 12, 64, 32, 2, 0.75, 512
DM = H * DH
BYTES_K = 4 * T * P * D                                            # 25 165 824
ROWS, FRAMES_IN_ROWS = 64, 255                                     # new rows per chunk / frames they cover at (T, N, tau) above
BYTES_POOL_ROWS_ONLY = 4 * FRAMES_IN_ROWS * P * D + 4 * ROWS * D   # pool_rows2_kernel<.., false> (infv_ltm_pool_rows): read the covered frames of k, write R
BYTES_POOL_PER_CHUNK = BYTES_POOL_ROWS_ONLY + 3 * 2 * ROWS * D     # ... in the pipeline (round 5) it also writes the rows' three bf16 planes for the projection GEMM
BYTES_LAYER = 4 * 2 * N * D + 4 * 2 * (D * DM + DM) + 4 * 2 * Q * DM + 4 * 2 * H * Q * N
BYTES_PER_CHUNK = BYTES_K + L * BYTES_LAYER                        # 39 727 104


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed passes over the video; default 200 x the number of GPUs (a pass takes ~18 ms on one GPU and "
                         "shrinks with the shard: the timed region stays >= 3 s)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--chunks", type=int, default=2048, help="chunks of the synthetic video (whole job)")
    ap.add_argument("--batch-chunks", type=int, default=42,
                    help="largest sub-batch (the library uses 32 for calls shorter than 768 chunks, e.g. multi-GPU shards)")
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="budget of the CPU baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-encode-video", action="store_true", help="skip the secondary per-chunk Q-former leg")
    ap.add_argument("--no-selfcheck", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary lines (split-bf16 V' projection, bf16 frame tokens)")
    ap.add_argument("--stub-engine", action="store_true",
                    help="TEST HOOK (tests/test_sharding_cpu.py): run the rank body on the CPU over gloo with a stand-in engine "
                         "-- launch, sharding, collective, max-over-ranks clock and the one JSON line, no GPU, no numbers worth reading")
    return ap.parse_args()


def spawn_ranks(n: int) -> int:
    """--gpus N without a torchrun environment: start N ranks as a child job.  Nothing here may initialise the
    GPU (the children own it; a process that has touched HIP must not exec or fork GPU users on this pool)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


POOL_THREADS_PER_CHUNK = 768 * 64          # pool_rows2_kernel: one 768-thread workgroup per (chunk, new row), 64 new rows per chunk (headline plan)


def pmc_traffic_per_full_launch():
    """HBM bytes of the pooling kernel's largest launch in the committed rocprofv3 PMC passes (profiles/*_pmc_summary.json: average
    FETCH_SIZE / WRITE_SIZE in KiB per dispatch) and the chunks that launch covered (its grid: since round 6 a call of 768+ chunks
    pools in ONE launch).  gfx950 correction: FETCH_SIZE counts the 128-B requests of a wide coalesced stream as 64 B, so it is
    doubled; WRITE_SIZE is exact (MI355X_MICROARCH.md, HBM).  Returns (bytes, source file, chunks of that launch)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
    if not files:
        return None, None, None
    d = json.load(open(files[-1]))
    grid_of = {}
    try:
        # the in-pipeline instantiation (512- or 1024-thread workgroups; the unroll factor is a tuning knob)
        def pick(t):
            # the in-pipeline kernel's FULL launch (keys carry the grid size; the PMC run's video has 42-chunk launches only,
            # the smaller grids are bench.py's own alone-leg calls)
            full = sorted((int(k.rsplit("grid=", 1)[1]), v) for k, v in d[t].items() if "pool_rows2_kernel<" in k and "grid=" in k)
            if full:
                grid_of[t] = full[-1][0]
                return full[-1][1][1]
            old = [v for k, v in d[t].items() if "pool_frames_kernel<" in k and (", 512" in k or ", 1024" in k)]   # rounds 1-2
            return old[0][1]
        fetch, write = pick("fetch"), pick("write")
    except (KeyError, IndexError):
        return None, None, None
    chunks = grid_of["fetch"] / POOL_THREADS_PER_CHUNK if "fetch" in grid_of else None
    return (2.0 * fetch + write) * 1024.0, os.path.basename(files[-1]), chunks


HBM_ACHIEVABLE_GBS = 6290.0    # what a pure streaming kernel reaches on this part (MI355X_MICROARCH.md, HBM section)


def pmc_fabric_bytes_per_chunk(batch_chunks: int):
    """Bytes that crossed the fabric (L2 <-> Infinity Cache / HBM) per chunk for EVERY kernel of the whole-video pipeline, from the
    committed PMC summary (full sub-batch launches only).  FETCH_SIZE undercounts 128-B requests of a wide coalesced stream by
    2x on gfx950 (MI355X_MICROARCH.md): the pooling stream is doubled (its traffic then equals its algorithmic bytes); for the
    other kernels, whose request mix is not known, the figure is given both ways (low: as counted, high: doubled).  Unlike the
    section-8d formula this charges what the memory system really moved: weights / B / scores that live in L2 cost nothing."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
    if not files:
        return None
    d = json.load(open(files[-1]))
    # (split3_rows_kernel ran once per sub-batch up to round 4; since round 5 only once per call, for the weights' planes: left out)
    frags = ("pool_rows2_kernel<", "gemm_x6_wide_kernel", "gemm_nt_lw_kernel", "chain_batch3_kernel<", "alpha_rows2_kernel<", "uc_fast_kernel<")
    per = {}
    pool_chunks = None
    try:
        for frag in frags:
            best = None
            for k_, v in d["fetch"].items():
                if frag in k_ and "grid=" in k_:
                    g = int(k_.rsplit("grid=", 1)[1])
                    if best is None or (v[0], g) > (best[0], best[1]):     # the most frequent launch shape = the full sub-batch
                        best = (v[0], g, k_)
            if best is None:
                continue
            key = best[2]
            f, w = d["fetch"][key][1] * 1024.0, d["write"].get(key, [0, 0.0])[1] * 1024.0
            pool = frag.startswith("pool_rows2")
            if pool:
                # (its LARGEST launch: since round 6 one launch pools a whole call of 768+ chunks; the per-sub-batch launches of a
                #  shorter video in the same summary are the alone leg's)
                big = max((int(k_.rsplit("grid=", 1)[1]), k_) for k_ in d["fetch"] if frag in k_ and "grid=" in k_)
                key = big[1]
                f, w = d["fetch"][key][1] * 1024.0, d["write"].get(key, [0, 0.0])[1] * 1024.0
                pool_chunks = big[0] / POOL_THREADS_PER_CHUNK
            per[frag.rstrip("<")] = ((2.0 * f if pool else f) + w, 2.0 * f + w)
    except (KeyError, IndexError, ValueError):
        return None
    if "pool_rows2_kernel" not in per:
        return None
    def per_chunk(name, v):                                   # the pooling launch covers pool_chunks chunks, every other kernel one sub-batch
        return v / (pool_chunks if name == "pool_rows2_kernel" and pool_chunks else float(batch_chunks))
    lo = sum(per_chunk(n_, v[0]) for n_, v in per.items())
    hi = sum(per_chunk(n_, v[1]) for n_, v in per.items())
    return {"low": lo, "high": hi, "source": os.path.basename(files[-1]),
            "per_kernel_high": {n_: round(per_chunk(n_, v[1])) for n_, v in per.items()}}


class _StubEngine:
    """Stand-in for LTMEngine in --stub-engine runs: same surface (consolidate / export_state / sync), trivial arithmetic on the CPU.
    It exists so that the multi-rank plumbing of this file can be exercised where there is no GPU."""

    def __init__(self):
        import torch
        self.L, self.N, self.d, self.dm = L, N, D, DM
        self._B = [torch.zeros(N, D) for _ in range(L)]

    def consolidate(self, k, q, projs, u, new_doc=True):
        import torch
        m = k.mean(dim=1)                                                # [C, D]
        for l in range(L):
            self._B[l] = m.mean(0, keepdim=True).expand(N, D).contiguous() * (l + 1)
        return m[:, None, None, :DM].expand(k.shape[0], L, Q, DM).contiguous()

    def export_state(self, l):
        import torch
        return self._B[l], torch.full((127,), 1.0 / 127)

    def sync(self):
        pass


def pmc_mfma_busy(batch_chunks: int = 42):
    """MFMA-pipe busy share of the projection GEMM (the path's MFMA kernel) and of the UC kernel's read-out from the committed
    rocprofv3 PMC pass (profiles/*_pmc_mfma_ltm.json: SQ_VALU_MFMA_BUSY_CYCLES against the kernel's busy cycles, tools/pmc_mfma.sh).
    Like ``traffic`` it is read from a committed profile of the same code, not collected live (counters need their own run)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_mfma_ltm.json")))
    if not files:
        return None
    d = json.load(open(files[-1]))
    out = {"source": os.path.basename(files[-1])}
    for key, frags in (("projection_gemm", ("gemm_x6_wide_kernel", "gemm_nt_lw_kernel")), ("uc_readout", ("uc_fast_kernel",))):
        for frag in frags:                                   # (the default GEMM first; the fp32-MFMA one in profiles of rounds 1-3)
            hit = [v for k_, v in d.items() if frag in k_ and "mfma_util_pct" in v]
            if hit:
                out[key] = round(float(hit[0]["mfma_util_pct"]), 1)
                out[key + "_kernel"] = frag
                if frag == "gemm_x6_wide_kernel" and batch_chunks == 42:
                    # the counter quotient is per chip (1024 SIMDs); this GEMM runs 63 workgroups (7 x 9 tiles of 384 x 256 per
                    # 42-chunk sub-batch), one per CU, beside the other kernels of the pipeline: busy share of the CUs it occupies
                    out[key + "_on_its_63_cus"] = round(float(hit[0]["mfma_util_pct"]) * 256.0 / 63.0, 1)
                break
    return out


# ----------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1): the oracle on the host cores, 1 thread and the best of a thread sweep
# ----------------------------------------------------------------------------------------------------------
def _thread_counts():
    """Thread counts of the sweep.  All cores is only tried up to 64: on the 256-thread GPU host one chunk of the
    reference-shaped port took 130 s with 256 ATen threads (0.008 chunks/s, oversubscription) against 0.6 s with 8."""
    import torch
    full = max(1, os.cpu_count() or 1, torch.get_num_threads())
    return sorted({n for n in (8, 16, 32, min(full, 64)) if n <= full} | {min(full, 8)})


def _time_dense(threads: int, budget_s: float, max_chunks: int):
    """Steady-state sticky chunks of the headline shape through oracle.DenseOracle (the reference's ATen sequence,
    including the four extra compute_probability calls of its density side effect, LTM.py:320-341)."""
    import torch
    from infinite_video_amd import synth
    from oracle.ltm_oracle import DenseOracle
    torch.set_num_threads(threads)
    layers = []
    for l in range(L):
        wk, bk, wv, bv = synth.layer_projections(l, D, DM)
        pk, pv = torch.nn.Linear(D, DM), torch.nn.Linear(D, DM)
        with torch.no_grad():
            pk.weight.copy_(torch.from_numpy(wk)); pk.bias.copy_(torch.from_numpy(bk))
            pv.weight.copy_(torch.from_numpy(wv)); pv.bias.copy_(torch.from_numpy(bv))
        layers.append(DenseOracle(N, H, DH, TAU, True, pk, pv, density_side_effect=True))
    qs = [torch.from_numpy(synth.layer_query(l, Q, DM)).unsqueeze(0) for l in range(L)]
    torch.manual_seed(synth.SEED_U)
    done, elapsed, c = 0, 0.0, 0
    with torch.no_grad():
        while True:
            k = torch.from_numpy(synth.frame_tokens(c, T, P, D)).unsqueeze(0)
            t0 = time.perf_counter()
            for l in range(L):
                layers[l].forward(k, qs[l], new_doc=(c == 0))
            dt = time.perf_counter() - t0
            if c > 0:                      # chunk 0 is the new-document warm-up (no sticky step)
                done += 1
                elapsed += dt
            c += 1
            if (elapsed >= budget_s and done >= 1) or done >= max_chunks:
                break
    return done, elapsed


def _time_closed(threads: int, budget_s: float, max_chunks: int):
    import torch
    from infinite_video_amd import synth
    from oracle.ltm_oracle import ClosedFormOracle
    torch.set_num_threads(threads)
    ws = [synth.layer_projections(l, D, DM) for l in range(L)]
    orcs = [ClosedFormOracle(N, H, DH, TAU, True, *ws[l], tokens_per_frame=P) for l in range(L)]
    qs = [synth.layer_query(l, Q, DM) for l in range(L)]
    u = synth.gibbs_uniforms(max_chunks + 1, L)
    done, elapsed, c = 0, 0.0, 0
    while True:
        k = synth.frame_tokens(c, T, P, D)
        t0 = time.perf_counter()
        for l in range(L):
            orcs[l].step(k, qs[l], new_doc=(c == 0), u=u[c, l])
        dt = time.perf_counter() - t0
        if c > 0:
            done += 1
            elapsed += dt
        c += 1
        if (elapsed >= budget_s and done >= 2) or done >= max_chunks:
            break
    return done, elapsed


def cpu_baselines(budget_s: float):
    """(cpu_baseline, cpu_closed_form): reference-shaped port and closed-form port, each at 1 thread and at the
    best of {8, 16, 32, all} threads (an oversubscribed all-cores run is not assumed to be the fastest)."""
    import torch
    saved = torch.get_num_threads()
    counts = _thread_counts()
    share = budget_s * 0.8 / (len(counts) + 2)            # the 1-thread run gets a double share
    dense = {}
    n1, t1 = _time_dense(1, 2 * share, 2)
    dense[1] = (n1, t1)
    for n in counts:
        if n != 1:
            dense[n] = _time_dense(n, share, 4)
    best = max(dense, key=lambda n: dense[n][0] / dense[n][1])
    total = sum(t for _, t in dense.values())
    base = {"value": dense[best][0] / dense[best][1], "unit": "frame-chunks/s", "cores": best, "kind": "port",
            "one_thread": dense[1][0] / dense[1][1],
            "threads_tried": {str(n): round(d / t, 4) for n, (d, t) in sorted(dense.items())},
            "host_cpus": os.cpu_count(),
            "sample": f"steady-state sticky chunks (T=256,N=256,2 layers sharing k) after 1 new-document chunk, "
                      f"oracle.DenseOracle = the reference's ATen sequence incl. its density side effect "
                      f"(LTM.py:320-341), {sum(d for d, _ in dense.values())} chunks over "
                      f"{len(dense)} thread counts, {total:.1f}s; value = best thread count"}
    closed = {}
    cshare = budget_s * 0.2 / (len(counts) + 1)
    for n in [1] + [n for n in counts if n != 1]:
        closed[n] = _time_closed(n, cshare, 24)
    cbest = max(closed, key=lambda n: closed[n][0] / closed[n][1])
    cf = {"value": closed[cbest][0] / closed[cbest][1], "unit": "frame-chunks/s", "cores": cbest,
          "kind": "port (closed form)", "one_thread": closed[1][0] / closed[1][1],
          "threads_tried": {str(n): round(d / t, 3) for n, (d, t) in sorted(closed.items())},
          "sample": f"steady-state sticky chunks, oracle.ClosedFormOracle (numpy/torch CPU), "
                    f"{sum(t for _, t in closed.values()):.1f}s"}
    torch.set_num_threads(saved)
    return base, cf


def module_forward_us(dev, k):
    """Per-call latency of the drop-in ``LongTermAttention`` module (the reference's operator surface, long_term_attention_gibbs.py:
    288-346 called from Qformer.py:221) at the headline shape: steady-state sticky calls on one layer, host issue and end-to-end
    time per call, median of five blocks of 200 calls; plus the kernel launches per call."""
    import torch
    from infinite_video_amd import _lib as _libmod
    from infinite_video_amd import synth
    from infinite_video_amd.long_term_attention_gibbs import LongTermAttention
    wk, bk, wv, bv = synth.layer_projections(0, D, DM)
    pk, pv = torch.nn.Linear(D, DM), torch.nn.Linear(D, DM)
    with torch.no_grad():
        pk.weight.copy_(torch.from_numpy(wk)); pk.bias.copy_(torch.from_numpy(bk))
        pv.weight.copy_(torch.from_numpy(wv)); pv.bias.copy_(torch.from_numpy(bv))
    m = LongTermAttention(head_size=DH, length=D, target_len=D, attn_func="softmax", attn_num_basis=N, continuous=True,
                          attn_drop=0.1, infinite_memory=True, n_layers=L, n_heads=H, affines=True, mask=True, mask_type="cnn",
                          kl_regularizer=False, proj_key=pk.to(dev), proj_value=pv.to(dev), sigma_0=None, mu_0=None,
                          sticky_memories=True, sigmas=None, tau=TAU, d_model=DM)
    ks = [k[c].unsqueeze(0) for c in range(min(8, k.shape[0]))]
    qq = torch.randn(1, Q, DM, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
    torch.manual_seed(0)
    for c in range(300):                   # one-time costs of a fresh module (code objects, pinned ring, clock ramp) stay outside
        m(ks[c % len(ks)], qq, new_doc=(c == 0), layer_n=0)
    torch.cuda.synchronize()
    n, res = 200, []
    launches0 = int(_libmod.load().infv_ltm_launch_count())
    for _ in range(5):
        t0 = time.perf_counter()
        for c in range(n):
            m(ks[c % len(ks)], qq, new_doc=False, layer_n=0)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0, t1 - t0))
    launches = (int(_libmod.load().infv_ltm_launch_count()) - launches0) / (5 * n)
    res.sort()
    e2e, host = res[len(res) // 2]
    return {"what": "LongTermAttention.forward (drop-in module), T=256 N=256 Q=32, steady-state sticky calls, one layer",
            "end_to_end_us": 1e6 * e2e / n, "host_issue_us": 1e6 * host / n, "best_block_us": 1e6 * res[0][0] / n,
            "launches_per_call": launches, "layer_steps_per_s": n / e2e}


# ----------------------------------------------------------------------------------------------------------
def selfcheck(eng_cls, dev, k, q, projs, u, ctx_timed, trace, batch_chunks):
    """Compare the timed path (one consolidate call over the whole block) with the per-chunk forward chain.
    (1) first 24 chunks from scratch: forward() per chunk, its draw forced to the timed run's recorded bins so that a
        uniform within rounding of a cdf edge cannot send the two chains apart; the per-chunk path's own draw is
        compared bin by bin (flips) and its ctx with the timed run's.
    (2) last chunk: consolidate the first C-1 chunks on a second engine, then ONE per-chunk forward() for chunk C-1."""
    import numpy as np
    import torch
    bins_all, _ = trace
    c_local = k.shape[0]
    n_head = min(24, c_local)
    out = {"head_chunks": n_head}
    ref = eng_cls(N, H, DH, D, P, tau=TAU, sticky=True, n_layers=L, max_q=Q, device=dev, max_batch_chunks=batch_chunks)
    worst, flips = 0.0, 0
    bins_host = bins_all[:n_head].cpu().numpy()
    for c in range(n_head):
        if c > 0:
            for l in range(L):
                ref.set_bins(l, bins_host[c, l])
        y = ref.forward(k[c], q, projs, u[c], new_doc=(c == 0))
        worst = max(worst, float((y - ctx_timed[c]).abs().max()))
        if c > 0:
            for l in range(L):
                flips += int((ref.last_draw(l)[0] != bins_host[c, l]).sum())
    out["head_max_abs_err"] = worst
    out["head_draw_flips"] = flips
    # (1b) the WHOLE call, chunk by chunk, the same way (the per-chunk chain goes on from chunk n_head): every draw of the timed call
    # against the per-chunk path's own draw -- the census tests/test_timed_path_gpu.py asserts a budget on, recorded here
    if c_local > n_head:
        bins_rest = bins_all[n_head:].cpu().numpy()
        w_all, f_all = worst, flips
        for c in range(n_head, c_local):
            for l in range(L):
                ref.set_bins(l, bins_rest[c - n_head, l])
            y = ref.forward(k[c], q, projs, u[c], new_doc=False)
            w_all = max(w_all, float((y - ctx_timed[c]).abs().max()))
            for l in range(L):
                f_all += int((ref.last_draw(l)[0] != bins_rest[c - n_head, l]).sum())
        out["whole_call_chunks"] = c_local
        out["whole_call_draws"] = (c_local - 1) * L * 512
        out["whole_call_draw_flips"] = f_all
        out["whole_call_max_abs_err"] = w_all
        out["whole_call_flip_budget"] = max(4, int(4e-5 * c_local * L * 512))
        worst = max(worst, w_all)
    if c_local >= 2:
        ref.consolidate(k[:c_local - 1], q, projs, u[:c_local - 1], new_doc=True)
        last_bins = bins_all[c_local - 1].cpu().numpy()
        for l in range(L):
            ref.set_bins(l, last_bins[l])
        y = ref.forward(k[c_local - 1], q, projs, u[c_local - 1], new_doc=False)
        ref.sync()
        out["last_chunk_max_abs_err"] = float((y - ctx_timed[c_local - 1]).abs().max())
        out["last_chunk_draw_flips"] = int(sum((ref.last_draw(l)[0] != last_bins[l]).sum() for l in range(L)))
        worst = max(worst, out["last_chunk_max_abs_err"])
    if c_local >= 2:
        # (3) the CPU oracle on the LAST chunk (checker only, outside every timed region): seeded with the HIP memory and scores
        # after chunk C-2 (`ref` was consolidated up to there above, before its forward of chunk C-1 -- so re-derive that state
        # from a third engine), fed the timed run's traced bins; its context must equal the timed call's last chunk
        from infinite_video_amd import synth
        from oracle.ltm_oracle import ClosedFormOracle
        pre = eng_cls(N, H, DH, D, P, tau=TAU, sticky=True, n_layers=L, max_q=Q, device=dev, max_batch_chunks=batch_chunks)
        pre.consolidate(k[:c_local - 1], q, projs, u[:c_local - 1], new_doc=True)
        pre.sync()
        kc = k[c_local - 1].cpu().numpy()
        o_worst = 0.0
        for l in range(L):
            ws = tuple(t.cpu().numpy() for t in projs[l])
            orc = ClosedFormOracle(N, H, DH, TAU, True, *ws, tokens_per_frame=P)
            orc.B_past = pre.export_state(l)[0].cpu().numpy().copy()
            orc.S_prev = np.asarray(pre.last_scores(l, Q), dtype=np.float32).copy()
            y = orc.step(kc, q[l].cpu().numpy(), new_doc=False, u=u[c_local - 1, l].cpu().numpy(),
                         bins_override=bins_all[c_local - 1, l].cpu().numpy())
            o_worst = max(o_worst, float(np.abs(y - ctx_timed[c_local - 1, l].cpu().numpy()).max()))
        out["last_chunk_vs_cpu_oracle_max_abs_err"] = o_worst
        worst = max(worst, o_worst)
        del pre
    out["max_abs_err"] = worst
    # gate: the north star's fp32 budget is 1e-3 (the tests hold the path to 1e-4); a handful of adjacent-bin flips between
    # the two HIP paths is expected over millions of draws (different fp32 association of the probabilities)
    flips_total = out["head_draw_flips"] + out.get("last_chunk_draw_flips", 0)
    out["ok"] = bool(worst <= SELFCHECK_MAX_ABS_ERR and flips_total <= SELFCHECK_MAX_FLIPS and
                     out.get("whole_call_draw_flips", 0) <= out.get("whole_call_flip_budget", 0))
    del ref
    return out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    # stdout carries exactly ONE line, the JSON of rank 0.  Libraries write there too (RCCL prints a version banner into
    # the C-level stdout buffer, which is flushed at exit, i.e. AFTER a Python print): point fd 1 at stderr for the whole
    # run and keep the real stdout for the final line.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.steps is None:
        args.steps = 200 * world
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    stub = args.stub_engine
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if stub:
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if not stub and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the LTM path has no CPU fallback")
    dev = torch.device("cpu") if stub else torch.device("cuda", local_rank)
    if not stub:
        torch.cuda.set_device(dev)

    from infinite_video_amd import synth
    from infinite_video_amd.video_memory import consolidate_video, shard_range

    start, stop = shard_range(args.chunks, world, rank)
    c_local = stop - start
    if stub:
        LTMEngine = None
        eng = _StubEngine()
    else:
        from infinite_video_amd.engine import LTMEngine
        eng = LTMEngine(N, H, DH, D, P, tau=TAU, sticky=True, n_layers=L, max_q=Q, device=dev,
                        max_batch_chunks=args.batch_chunks)
    projs = [tuple(torch.from_numpy(a).to(dev) for a in synth.layer_projections(l, D, DM)) for l in range(L)]
    q = torch.from_numpy(np.stack([synth.layer_query(l, Q, DM) for l in range(L)])).to(dev)
    u = torch.from_numpy(synth.gibbs_uniforms(args.chunks, L)[start:stop]).to(dev)
    # synthetic frame tokens ~ N(0,1), distinct per chunk (51.5 GB at 2048 chunks: far beyond the
    # 256 MiB Infinity Cache, so the pool really streams from HBM)
    k = torch.empty(c_local, T * P, D, device=dev, dtype=torch.float32)
    gen = torch.Generator(device=dev).manual_seed(synth.SEED_K + start)
    for i in range(0, c_local, 64):
        k[i:i + 64].normal_(generator=gen)
    dev_sync = (lambda: None) if stub else torch.cuda.synchronize
    dev_sync()

    tim = {}                                                        # shard / all-gather decomposition of the timed steps

    def one_step(timings=None):
        return consolidate_video(eng, k, q, projs, u, timings=timings)

    def fence():
        dev_sync()
        if world > 1:
            dist.barrier()
        dev_sync()

    for _ in range(args.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx, mem = one_step()
    fence()
    elapsed = time.perf_counter() - t0
    # shard / all-gather split of a step: a separate short pass (its extra device-wide sync behind the collective is not part of
    # the timed region above, so a multi-GPU run overlaps step i's all-gather with step i+1's consolidation as a caller would)
    for _ in range(min(5, args.steps)):
        one_step(tim)
    fence()
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert bool(torch.isfinite(ctx).all()), "non-finite consolidation output"
    if stub:
        # the plumbing line of a --stub-engine run: same keys as the real one where they exist without a GPU
        if rank == 0:
            out = {"metric": "frame-chunks/sec consolidated (max_int=256, num_basis=256, d=768)", "stub_engine": True,
                   "value": args.chunks * args.steps / elapsed, "unit": "frame-chunks/s", "n_gpus": world, "steps": args.steps,
                   "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
                   "config": {"chunks": args.chunks, "chunks_per_gpu": c_local},
                   "gathered_counts": [float(x) for x in mem.count],
                   "shard_ms": 1e3 * tim["shard_s"] / tim["calls"], "allgather_ms": 1e3 * tim["allgather_s"] / tim["calls"]}
            real_stdout.write(json.dumps(out) + "\n")
            real_stdout.flush()
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- self-check of the code path that was just timed (same engine, same call, plus a draw trace) ----
    check, bins_default = None, None
    if not args.no_selfcheck:
        trace = eng.set_trace(c_local)
        ctx_chk, _ = one_step()
        eng.set_trace(0)
        assert torch.equal(ctx_chk, ctx), "the consolidation is not reproducible run to run"
        check = selfcheck(LTMEngine, dev, k, q, projs, u, ctx_chk, trace, args.batch_chunks)
        bins_default = trace[0].clone()
        del trace

    def draw_flips(engine, tokens):
        """Draws of `engine`'s call over the same video that differ from the default line's (both free-running): how many of the
        call's (C-1) x L x S draws, how many chunks hold at least one, and the largest bin distance."""
        if bins_default is None:
            return None
        tr = engine.set_trace(c_local)
        consolidate_video(engine, tokens, q, projs, u)
        torch.cuda.synchronize()
        engine.set_trace(0)
        diff = tr[0] != bins_default
        dist_max = int((tr[0] - bins_default).abs().max()) if bool(diff.any()) else 0
        return {"draws_differing": int(diff.sum()), "of": int((c_local - 1) * L * S),
                "chunks_with_a_differing_draw": int(diff.flatten(1).any(1).sum()), "max_bin_distance": dist_max}

    # ---- secondary line (N = 1): the same call with the V' half of the projection as three bf16 MFMA products
    #      (INFV_VPROJ_SPLIT=1 at engine creation).  V' only feeds the read-out (1e-3 budget); the draw is unchanged. ----
    vsplit = None
    if world == 1 and not args.no_secondary and os.environ.get("INFV_VPROJ_SPLIT", "0") in ("", "0"):
        os.environ["INFV_VPROJ_SPLIT"] = "1"
        try:
            eng2 = LTMEngine(N, H, DH, D, P, tau=TAU, sticky=True, n_layers=L, max_q=Q, device=dev,
                             max_batch_chunks=args.batch_chunks)
        finally:
            del os.environ["INFV_VPROJ_SPLIT"]
        for _ in range(2):
            consolidate_video(eng2, k, q, projs, u)
        torch.cuda.synchronize()
        n2 = max(1, min(args.steps, 40))
        t1 = time.perf_counter()
        for _ in range(n2):
            ctx2, _ = consolidate_video(eng2, k, q, projs, u)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        vsplit = {"dtype": "f32 (V' projection bf16x3)", "value": args.chunks * n2 / dt2, "unit": "frame-chunks/s",
                  "steps": n2, "ms_per_step": 1e3 * dt2 / n2, "max_abs_diff_vs_f32": float((ctx2 - ctx).abs().max()),
                  "draw_flips": draw_flips(eng2, k)}
        del eng2, ctx2

    # ---- secondary line (N = 1): the projection GEMM on the fp32 MFMA pipe (INFV_PROJ_X6=0 at engine creation: gemm_nt_lw_kernel,
    #      the default of rounds 1-3).  The default since round 4 carries both operands as exact three-piece bf16 splits and
    #      accumulates their six partial products in fp32 on the bf16 MFMA pipe (gemm_x6_wide_kernel; against fp64 its error is
    #      below the fp32-MFMA GEMM's, tests/test_ltm_gpu.py) ----
    proj_f32mfma = None
    if world == 1 and not args.no_secondary and os.environ.get("INFV_PROJ_X6", "") == "":
        os.environ["INFV_PROJ_X6"] = "0"
        try:
            eng4 = LTMEngine(N, H, DH, D, P, tau=TAU, sticky=True, n_layers=L, max_q=Q, device=dev,
                             max_batch_chunks=args.batch_chunks)
        finally:
            del os.environ["INFV_PROJ_X6"]
        for _ in range(2):
            consolidate_video(eng4, k, q, projs, u)
        torch.cuda.synchronize()
        n4 = max(1, min(args.steps, 40))
        t1 = time.perf_counter()
        for _ in range(n4):
            ctx4, _ = consolidate_video(eng4, k, q, projs, u)
        torch.cuda.synchronize()
        dt4 = time.perf_counter() - t1
        proj_f32mfma = {"dtype": "f32 MFMA (32x32x2) projection GEMM instead of the six-product bf16 form",
                        "value": args.chunks * n4 / dt4, "unit": "frame-chunks/s", "steps": n4, "ms_per_step": 1e3 * dt4 / n4,
                        "max_abs_diff_vs_default": float((ctx4 - ctx).abs().max()), "draw_flips": draw_flips(eng4, k)}
        del eng4, ctx4

    # ---- secondary line (N = 1): the optional bf16 producer layout of the frame tokens (infv_ltm_set_token_dtype; half the
    #      bytes of the only HBM-heavy stream; everything after the pooling stays fp32) ----
    bf16_tokens = None
    if world == 1 and not args.no_secondary:
        k16 = torch.empty(c_local, T * P, D, device=dev, dtype=torch.bfloat16)
        for i in range(0, c_local, 64):
            k16[i:i + 64] = k[i:i + 64]
        eng3 = LTMEngine(N, H, DH, D, P, tau=TAU, sticky=True, n_layers=L, max_q=Q, device=dev,
                         max_batch_chunks=args.batch_chunks)
        for _ in range(2):
            consolidate_video(eng3, k16, q, projs, u)
        torch.cuda.synchronize()
        n3 = max(1, min(args.steps, 40))
        t1 = time.perf_counter()
        for _ in range(n3):
            ctx3, _ = consolidate_video(eng3, k16, q, projs, u)
        torch.cuda.synchronize()
        dt3 = time.perf_counter() - t1
        bf16_tokens = {"dtype": "f32 arithmetic on bf16-rounded frame tokens", "value": args.chunks * n3 / dt3,
                       "unit": "frame-chunks/s", "steps": n3, "ms_per_step": 1e3 * dt3 / n3,
                       "max_abs_diff_vs_f32_tokens": float((ctx3 - ctx).abs().max()), "draw_flips": draw_flips(eng3, k16)}
        del eng3, ctx3, k16

    # ---- one multi-GPU shard on this GPU: a 256-chunk consolidate_video including the packing AND the collective: a
    #      world-of-one "nccl" (= RCCL) group is initialised for this leg, so all_gather_into_tensor really runs ----
    shard256_ms, shard256_rccl = None, False
    if world == 1 and c_local >= 256:
        own_group = False
        if not dist.is_initialized():
            try:
                with socket.socket() as s_:
                    s_.bind(("127.0.0.1", 0))
                    port = s_.getsockname()[1]
                os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
                dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
                own_group = True
            except Exception as exc:                      # the leg still runs, without the collective
                print(f"[bench] RCCL world-of-one group not available: {exc}", file=sys.stderr)
        shard256_rccl = dist.is_initialized()
        try:
            for _ in range(2):
                consolidate_video(eng, k[:256], q, projs, u[:256])
            torch.cuda.synchronize()
            ts = []
            for _ in range(7):
                t1 = time.perf_counter()
                consolidate_video(eng, k[:256], q, projs, u[:256])
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t1)
            shard256_ms = 1e3 * sorted(ts)[len(ts) // 2]
        finally:
            if own_group:
                dist.destroy_process_group()

    # ---- the HBM-bound kernel on its own (no other stream running): 5 launches of one sub-batch ----
    nb = min(args.batch_chunks if c_local >= 768 else min(args.batch_chunks, 32), c_local)
    plan = eng.ensure_plan(T)
    assert (len(plan.inf_row_box), int((plan.inf_row_end - plan.inf_row_begin).sum())) == (ROWS, FRAMES_IN_ROWS)
    eng.pool_rows(k[:nb])                                            # the same kernel, same launch size, nothing else running
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(5):
        eng.pool_rows(k[:nb])
    ev1.record()
    torch.cuda.synchronize()
    alone_gbs = 5 * nb * BYTES_POOL_ROWS_ONLY / (ev0.elapsed_time(ev1) * 1e-3) / 1e9

    # ---- roofline leg: one more pass with HIP events around every kernel launch ----
    eng.profile(True)
    eng.consolidate(k, q, projs, u, new_doc=True)
    prof = eng.profile_read()
    eng.profile(False)
    pool_n, pool_ms = prof["pool"]
    # (the planes are written for sub-batches of >= 1024 rows when the default six-product GEMM runs: every sub-batch of this call)
    planes = os.environ.get("INFV_PROJ_X6", "") != "0" and os.environ.get("INFV_VPROJ_SPLIT", "0") in ("", "0")
    per_chunk = BYTES_POOL_PER_CHUNK if planes else BYTES_POOL_ROWS_ONLY
    pool_bytes = c_local * per_chunk
    achieved = pool_bytes / (pool_ms * 1e-3) / 1e9 if pool_ms > 0 else 0.0
    traffic_pmc, traffic_src, traffic_chunks = pmc_traffic_per_full_launch()
    # chunks of the pooling kernel's main launch in THIS run: one launch for the rank's whole call when it has 768+ chunks (round 6),
    # a sub-batch otherwise; `achieved` = bytes of ALL pooling launches of the pass / their summed duration (HIP events on the launch stream)
    launch_chunks = (c_local - 1) if c_local >= 768 else nb
    traffic = traffic_pmc / traffic_chunks * launch_chunks if traffic_pmc and traffic_chunks else traffic_pmc
    roofline = {
        "kernel": "pool_rows2_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "traffic_from_committed_profile": True,
        "achieved_alone": alone_gbs, "frac_alone": alone_gbs / HBM_PEAK_GBS,
        "note": "achieved = in situ, while the pool shares the chip with the chain and update/read-out streams; "
                "achieved_alone = same launch size through infv_ltm_pool_rows, nothing else running",
        "launches": pool_n, "avg_launch_ms": pool_ms / max(pool_n, 1),
        "launch_chunks": launch_chunks, "bytes_per_full_launch": launch_chunks * per_chunk,
        "traffic_note": "PMC bytes (2 x FETCH_SIZE + WRITE_SIZE) of the pooling kernel's largest launch in the committed summary, per chunk, "
                        "x launch_chunks of this run's launch",
        "traffic_pmc_launch_chunks": traffic_chunks,
        "whole_path_frac": (args.chunks * args.steps / elapsed) * BYTES_PER_CHUNK / 1e9 / (HBM_PEAK_GBS * world),
        "kernel_ms_per_pass": {name: round(ms, 3) for name, (n, ms) in prof.items()},
        "mfma_busy_pct": pmc_mfma_busy(args.batch_chunks), "mfma_busy_from_committed_profile": True,
    }
    fabric = pmc_fabric_bytes_per_chunk(args.batch_chunks)
    if fabric is not None:
        cps = args.chunks * args.steps / elapsed / world            # chunks/s of ONE GPU
        roofline["fabric_bytes_per_chunk"] = fabric
        # the honest ceiling: what the memory system really moved per chunk (PMC) against what a streaming kernel can reach
        roofline["frac_of_achievable"] = {"low": cps * fabric["low"] / 1e9 / HBM_ACHIEVABLE_GBS,
                                          "high": cps * fabric["high"] / 1e9 / HBM_ACHIEVABLE_GBS,
                                          "achievable_gbs": HBM_ACHIEVABLE_GBS,
                                          "bytes_from_committed_profile": fabric["source"],
                                          "note": "THIS run's chunks/s x fabric_bytes_per_chunk of the committed PMC pass named in "
                                                  "bytes_from_committed_profile (the builder's run of the same pipeline: counters need a run of "
                                                  "their own; stale if the kernels changed since) / 6.29 TB/s; whole_path_frac uses the section-8d "
                                                  "formula (39.7 MB/chunk incl. L2-resident weights / B / scores) against the 8 TB/s spec peak"}

    # ---- secondary leg (rank 0, N = 1): the same chunk shape through the whole video Q-former (encode_video
    #      counterpart: short-term cross-attention + LTM + merge + query FFN + llama_proj), per-chunk calls ----
    encode_video = None
    if rank == 0 and world == 1 and not args.no_encode_video:
        from infinite_video_amd.video_qformer import InfVideoEncoder
        model = InfVideoEncoder(num_basis=N, tau=TAU, alpha=0.9, sticky=True)
        model.load_reference_state_dict(synth.video_qformer_weights())
        model = model.to(dev)
        n_enc = 48
        uu = torch.from_numpy(synth.gibbs_uniforms(n_enc + 2, L)).to(dev)
        for c in range(2):
            model.encode_frames(k[c % c_local].unsqueeze(0), new_video=(c == 0), u=uu[c])
        torch.cuda.synchronize()
        from infinite_video_amd import _lib as _libmod
        launches0 = int(_libmod.load().infv_ltm_launch_count())
        t1 = time.perf_counter()
        for c in range(n_enc):
            model.encode_frames(k[c % c_local].unsqueeze(0), new_video=False, u=uu[2 + c])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        launches_per_chunk = (int(_libmod.load().infv_ltm_launch_count()) - launches0) / n_enc
        flop = L * 2 * (2 * H * Q * D * T * P)                       # two [H*Q x d x T*P] contractions per layer
        encode_video = {"what": "per-chunk encode_video counterpart (2-layer video Q-former + LTM + llama_proj), alpha=0.9",
                        "dtype": "f32 (short-term attention contractions as 3 bf16 MFMA products, ~1e-5 relative; LTM exact f32)",
                        "chunks_per_s": n_enc / dt, "ms_per_chunk": 1e3 * dt / n_enc, "chunks": n_enc,
                        "launches_per_chunk": launches_per_chunk,
                        "short_attention_gflop_per_chunk": flop / 1e9,
                        "short_attention_tflops_over_whole_chunk_time": flop * n_enc / dt / 1e12}
        # the same model, layer-major over a whole video (infv_vqf_encode_video: frame tokens read once per chunk)
        n_lm = min(252, c_local)
        u_lm = torch.from_numpy(synth.gibbs_uniforms(n_lm, L)).to(dev)
        model.encode_frames_batch(k[:n_lm], new_video=True, u=u_lm)     # full-size warm-up: workspaces grow here
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t1 = time.perf_counter()
            model.encode_frames_batch(k[:n_lm], new_video=True, u=u_lm)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t1)
        dt = sorted(ts)[1]
        encode_video["layer_major"] = {"chunks": n_lm, "chunks_per_s": n_lm / dt, "ms_per_chunk": 1e3 * dt / n_lm,
                                       "short_attention_tflops_over_whole_chunk_time": flop * n_lm / dt / 1e12}
        # the same two schedules with exact-fp32 MFMA contractions (infv_vqf_set_precision(h, 1))
        model.exact_fp32 = True
        for c in range(2):
            model.encode_frames(k[c % c_local].unsqueeze(0), new_video=(c == 0), u=uu[c])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for c in range(n_enc):
            model.encode_frames(k[c % c_local].unsqueeze(0), new_video=False, u=uu[2 + c])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        model.encode_frames_batch(k[:n_lm], new_video=True, u=u_lm)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        model.encode_frames_batch(k[:n_lm], new_video=True, u=u_lm)
        torch.cuda.synchronize()
        dt_lm = time.perf_counter() - t1
        encode_video["exact_f32"] = {"dtype": "f32", "ms_per_chunk": 1e3 * dt / n_enc, "layer_major_ms_per_chunk": 1e3 * dt_lm / n_lm}
        del model

    # ---- the drop-in operator (what the unchanged reference drivers call, Qformer.py:216-223): LongTermAttention.forward per call ----
    module_forward = None
    if rank == 0 and world == 1 and not args.no_encode_video:
        module_forward = module_forward_us(dev, k)

    if rank == 0:
        value = args.chunks * args.steps / elapsed
        v_split = os.environ.get("INFV_VPROJ_SPLIT", "0") not in ("", "0")
        out = {
            "metric": "frame-chunks/sec consolidated (max_int=256, num_basis=256, d=768)",
            "value": value, "unit": "frame-chunks/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 (V' projection bf16x3)" if v_split else "f32", "data": "synthetic",
            # every value is carried at full f32 precision everywhere.  The new-row projection GEMM multiplies f32 operands as exact
            # three-piece bf16 splits (8 + 8 + 8 significand bits) -- six partial products, each exact, accumulated in f32 on the
            # bf16 MFMA pipe: error against f64 at or below the f32-MFMA GEMM's (tests/test_ltm_gpu.py), same goldens, same drawn
            # bins.  INFV_PROJ_X6=0 runs that GEMM as f32 MFMAs instead: timed below as secondary_proj_f32_mfma.
            "dtype_detail": {"tokens": "f32", "pooling": "f32", "state_and_scores": "f32", "draw": "f32 cdf vs f64 uniforms (torch CPU semantics)",
                             "projection_gemm": ("f32 MFMA" if os.environ.get("INFV_PROJ_X6", "") == "0" else
                                                 "bf16x3-split operands (exact 24-bit), 6 MFMA products, f32 accumulate"),
                             "value_projection": "bf16x3 (3 products)" if v_split else "same GEMM as the scores"},
            "dtype_note": ("projection GEMM: f32 MFMA (INFV_PROJ_X6=0)" if os.environ.get("INFV_PROJ_X6", "") == "0" else
                           "projection GEMM: f32 operands as exact 3-piece bf16 splits, 6 MFMA products, f32 accumulation (f32-accurate)"),
            "config": {"workload": f"{args.chunks}-chunk synthetic video, max_int=256 frames x 32 tokens x 768, "
                                   "num_basis=256, tau=0.75, sticky, 2 video-Q-former LTM layers, "
                                   "Q=32 queries, LLM/Q-former stubbed (BASELINE configs[1]/[2])",
                       "chunks": args.chunks, "chunks_per_gpu": c_local, "layer_steps_per_s": value * L,
                       "parallelism": f"chunk-block sharding x{world} + 1 all-gather of consolidated memory"},
            "roofline": roofline,
        }
        # every knob of the library / runtime that was set for this run (an empty list = the shipped defaults)
        out["env_overrides"] = sorted(f"{k_}={v_}" for k_, v_ in os.environ.items()
                                      if k_.startswith("INFV_") or k_ in ("GPU_MAX_HW_QUEUES", "HIP_VISIBLE_DEVICES"))
        if check is not None:
            out["selfcheck"] = check
            out["selfcheck_max_abs_err"] = check["max_abs_err"]
            out["selfcheck_ok"] = check["ok"]
        # decomposition of a step on this rank (host clocks around the sync in front of the collective and behind it)
        if tim.get("calls"):
            out["shard_ms"] = 1e3 * tim["shard_s"] / tim["calls"]
            out["allgather_ms"] = 1e3 * tim["allgather_s"] / tim["calls"]
        if shard256_ms is not None:
            out["shard256_ms"] = shard256_ms
            out["shard256_includes_rccl_all_gather"] = shard256_rccl
            # what the 1 -> 8 GPU curve can be at best: the whole video on one GPU against one 256-chunk shard (+ its all-gather)
            # the world-of-one collective inside shard256_ms is a no-op; an 8-rank all-gather of 1.77 MB per rank is not.  It cannot be
            # measured on one GPU (two ranks cannot share a device under RCCL), so the prediction carries a stated ASSUMPTION:
            # RCCL small-message latency (~20 us for a one-node all-gather launch + ring set-up) + 7 ring steps x 1.77 MB at ~64 GB/s per
            # link direction (xGMI: 7 links x ~153 GB/s bidirectional per GPU; ring collectives are per-link bound)
            payload_mb = 4.0 * (L * N * D + L * 127 + L * Q * DM + 1) / 1e6
            allgather_assumed = 0.020 + 7 * payload_mb / 64.0e3 * 1e3
            out["allgather_payload_mb_per_rank"] = payload_mb
            out["allgather_ms_assumed_8"] = allgather_assumed
            out["allgather_ms_assumed_8_note"] = ("assumed, not measured (one GPU): RCCL's RING all-gather on a fully connected xGMI node: "
                                                  "20 us launch/latency + 7 ring steps of the per-rank payload at 64 GB/s per link direction")
            # the same exchange as ONE hop (every rank sends its 1.77 MB straight to each of its 7 peers over 7 separate links:
            # SURVEY.md section 8e; what a direct / one-shot all-gather does for messages this small): 20 us + one payload per link
            allgather_one_hop = 0.020 + payload_mb / 64.0e3 * 1e3
            out["allgather_ms_one_hop_8"] = allgather_one_hop
            out["predicted_speedup_8_one_hop"] = (1e3 * elapsed / args.steps) / (shard256_ms + allgather_one_hop) if world == 1 else None
            out["predicted_speedup_8"] = (1e3 * elapsed / args.steps) / (shard256_ms + allgather_assumed) if world == 1 else None
            out["predicted_speedup_8_without_allgather"] = (1e3 * elapsed / args.steps) / shard256_ms if world == 1 else None
        if vsplit is not None:
            out["secondary_vproj_bf16x3"] = vsplit
        if proj_f32mfma is not None:
            out["secondary_proj_f32_mfma"] = proj_f32mfma
        if bf16_tokens is not None:
            out["secondary_bf16_tokens"] = bf16_tokens
        if encode_video is not None:
            out["encode_video"] = encode_video
        if module_forward is not None:
            out["forward_us"] = module_forward["end_to_end_us"]
            out["module_forward"] = module_forward
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], out["cpu_closed_form"] = cpu_baselines(args.cpu_seconds)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
            out["speedup_vs_cpu_closed_form"] = value / out["cpu_closed_form"]["value"]
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if world > 1:
        dist.destroy_process_group()
    if check is not None and not check["ok"]:
        raise SystemExit(f"bench.py: selfcheck failed ({check})")


if __name__ == "__main__":
    main()
