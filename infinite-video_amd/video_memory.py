"""Per-chunk inference loop and its multi-GPU sharding.

Single GPU: the loop of the reference eval drivers
(eval_code/eval/run_inference_inf_video_llama_nextqa.py:179-196 -- ``new_video`` true on the
first chunk only, one consolidation per chunk, mean of the per-chunk outputs at :194) with the
Q-former/LLM stubbed, i.e. the LTM operator called directly on each chunk's frame tokens.

Multi GPU: one process per GPU; a video of C chunks is cut into
contiguous blocks, rank r consolidates its block as its own document (``new_doc`` on its first
chunk, an independent sticky chain per layer), and ONE all-gather (RCCL over xGMI when the
process group is "nccl") exchanges each rank's consolidated memory -- B_past per layer, the
sticky bin masses, and the local SUM of per-chunk outputs with its chunk count -- before the
LLM forward.  The reference's final reduction is a plain mean over chunks and ``llama_proj`` is
linear (infinityqa.py:342), so summing before the gather is exact up to fp32 rounding.
There is no other collective on the data path.

The Gibbs uniforms are keyed by GLOBAL chunk index, so rank r's result equals a single-GPU
run on the sub-video made of rank r's chunks.

Correctness mode (``consolidate_video(..., handoff=True)``): the ranks hand the memory chain on from block to block
(point-to-point send/recv of ~5.5 MB at the headline shape) and reproduce the single-stream run -- bit for bit when every
rank uses the same q / projections and no block ends in a short sub-batch (see ``consolidate_video``); they run one after
the other.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_chunks: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block of chunk indices owned by ``rank`` (first ``n_chunks % world`` ranks get one more)."""
    if not 0 <= rank < world:
        raise ValueError("rank outside world")
    base, rem = divmod(n_chunks, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


@dataclass
class ConsolidatedMemory:
    """What the ranks exchange: index 0 of every tensor is the owning rank."""
    B: torch.Tensor          # [R, L, N, d]   coefficient matrices (B_past)
    bin_mass: torch.Tensor   # [R, L, 127]    unnormalised sticky bin masses
    ctx_sum: torch.Tensor    # [R, L, Q, dm]  sum over the rank's chunks of the per-chunk outputs
    count: torch.Tensor      # [R]            chunks consolidated by the rank
    scores: Optional[torch.Tensor] = None   # [R, L, H, Q, N]  bias-free scores of the rank's last step (``with_scores=True``)

    def mean_embedding(self) -> torch.Tensor:
        """Mean over ALL chunks of the video of the per-chunk outputs, [L, Q, dm]
        (the reference's torch.mean(torch.stack(video_embs)) at nextqa.py:194)."""
        return self.ctx_sum.sum(0) / self.count.sum()


def pack_local_memory(engine, ctx_local: torch.Tensor, with_scores: bool = False) -> torch.Tensor:
    """Flatten this rank's consolidated memory into one fp32 payload (one collective, not four).  ``with_scores``: also the last
    step's scores [L, H, Q, N] (SURVEY.md section 8e's full payload: +0.79 MB per rank at the headline shape)."""
    L = engine.L
    parts = []
    masses = []
    for l in range(L):
        B, mass = engine.export_state(l)
        parts.append(B.reshape(-1))
        masses.append(mass.reshape(-1))
    count = torch.full((1,), float(ctx_local.shape[0]), device=ctx_local.device, dtype=torch.float32)
    extra = [engine.last_scores_device(int(ctx_local.shape[2])).reshape(-1).to(torch.float32)] if with_scores else []
    return torch.cat(parts + masses + extra + [ctx_local.sum(0).reshape(-1), count])


def unpack_memory(payload: torch.Tensor, world: int, L: int, N: int, d: int, Q: int, dm: int, H: int = 0) -> ConsolidatedMemory:
    """``H`` > 0: the payload carries the last scores [L, H, Q, N] behind the bin masses (``pack_local_memory(with_scores=True)``)."""
    payload = payload.reshape(world, -1)
    nB, nM, nC, nS = L * N * d, L * 127, L * Q * dm, L * H * Q * N
    if payload.shape[1] != nB + nM + nS + nC + 1:
        raise ValueError("payload size does not match the memory layout")
    return ConsolidatedMemory(
        B=payload[:, :nB].reshape(world, L, N, d),
        bin_mass=payload[:, nB:nB + nM].reshape(world, L, 127),
        ctx_sum=payload[:, nB + nM + nS:nB + nM + nS + nC].reshape(world, L, Q, dm),
        count=payload[:, -1],
        scores=payload[:, nB + nM:nB + nM + nS].reshape(world, L, H, Q, N) if H > 0 else None)


def _device_collectives(group=None) -> bool:
    """True when the group's backend moves device tensors itself ("nccl" = RCCL over xGMI); gloo (CPU tests, two
    processes on one GPU) needs host staging."""
    return dist.get_backend(group) == "nccl"


def _all_gather_payload(payload: torch.Tensor, world: int, group=None) -> torch.Tensor:
    if _device_collectives(group):
        gathered = torch.empty(world * payload.numel(), device=payload.device, dtype=payload.dtype)
        dist.all_gather_into_tensor(gathered, payload, group=group)
        return gathered
    host = payload.cpu()
    parts = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(parts, host, group=group)
    return torch.cat(parts).to(payload.device)


def _global_rank(group, group_rank: int) -> int:
    """torch.distributed's send / recv take GLOBAL ranks; the hand-off chain is numbered inside ``group``."""
    return group_rank if group is None else dist.get_global_rank(group, group_rank)


def _send(t: torch.Tensor, dst: int, group=None):
    dist.send(t if _device_collectives(group) else t.cpu(), _global_rank(group, dst), group=group)


def _recv(t: torch.Tensor, src: int, group=None) -> torch.Tensor:
    if _device_collectives(group):
        dist.recv(t, _global_rank(group, src), group=group)
        return t
    host = torch.empty(t.shape, dtype=t.dtype)
    dist.recv(host, _global_rank(group, src), group=group)
    return host.to(t.device)


def consolidate_video(engine, k_local: torch.Tensor, q: torch.Tensor, projs: Sequence,
                      u_local: Optional[torch.Tensor], group=None, handoff: bool = False,
                      timings: Optional[dict] = None, with_scores: bool = False) -> Tuple[torch.Tensor, ConsolidatedMemory]:
    """Consolidate this rank's block of chunks and all-gather the consolidated memory.

    k_local [C_local, T*P, d]; q [L, Q, dm]; u_local [C_local, L, S] (rows keyed by global chunk id).
    Returns (per-chunk outputs of this rank [C_local, L, Q, dm], gathered ConsolidatedMemory).
    Works without an initialised process group (world = 1, no collective); with an initialised group the
    all-gather is issued whatever its size (a world of one still goes through RCCL).

    ``handoff=False`` (default, the throughput mode of SURVEY.md section 8e): every rank's block is its own document;
    the ranks run concurrently and the result equals single-GPU runs on the sub-videos -- NOT the single-stream run
    (a block starts from an empty memory; the difference decays within ~63 chunks of a boundary).
    ``handoff=True`` (the correctness mode): rank r first receives rank r-1's chain state (B, projected memory, scores,
    sticky bin masses: ``engine.export_chain_state``), imports it and runs its block with ``new_doc=False``, then
    passes its own state on to rank r+1.  Every chunk then sees the memory of all earlier chunks exactly as in the
    reference's one-process loop (long_term_attention_gibbs.py:194-222): the outputs equal the single-stream run bit
    for bit -- and the ranks run one after the other, so it does not scale.  Bit for bit holds when (a) every rank passes
    the same ``q`` and ``projs`` (the blob carries the scores under the exporter's query; nothing checks this) and (b) no
    block but the last ends in a sub-batch of fewer than 1024 new rows (16 chunks at the headline shape): such a tail takes
    the split-K form of the projection, whose partial sums are ordered differently -- same values to fp32 rounding, a draw
    may then differ by an index (tests/test_sharding_gpu.py cuts 129 chunks as 65 + 64 for this reason).  The blob has a
    header (shape of the exporting handle); importing one of another shape fails at the next library call.

    This is the hand-over point to the LLM forward, so it waits for the consolidation (``engine.sync()``) and
    raises if the persistent chain kernel reported a failure instead of passing an invalid memory on.

    ``with_scores``: the gathered memory also carries every rank's last scores [L, H, Q, N] (``ConsolidatedMemory.scores``) --
    SURVEY.md section 8e's full payload; off by default (B and the 127 bin masses are enough to continue a sticky draw).

    ``timings`` (a dict, benchmarks only): accumulates ``shard_s`` (this rank's consolidation + packing, up to the sync in
    front of the collective) and ``allgather_s`` (the collective, synchronised) so that a multi-GPU line decomposes."""
    import time
    t_begin = time.perf_counter() if timings is not None else 0.0
    have_group = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if have_group else 1
    rank = dist.get_rank(group) if have_group else 0
    Q = int(q.shape[1])
    chained = handoff and world > 1
    if chained and rank > 0:
        blob = torch.empty(engine.chain_state_numel(Q), device=k_local.device, dtype=torch.float32)
        blob = _recv(blob, rank - 1, group)
        engine.import_chain_state(Q, blob)
    ctx = engine.consolidate(k_local, q, projs, u_local, new_doc=not (chained and rank > 0))
    if chained and rank < world - 1:
        blob = engine.export_chain_state(Q)
        engine.sync()                                # complete (and free of a latched chain failure) before it leaves
        _send(blob, rank + 1, group)
    payload = pack_local_memory(engine, ctx, with_scores)   # stream-ordered behind the consolidation: no host round trip in between
    engine.sync()                                    # a latched chain failure raises here, before anything is handed on
    t_shard = time.perf_counter() if timings is not None else 0.0
    gathered = _all_gather_payload(payload, world, group) if have_group else payload
    if timings is not None:
        if have_group and payload.is_cuda:
            torch.cuda.synchronize(payload.device)
        timings["shard_s"] = timings.get("shard_s", 0.0) + (t_shard - t_begin)
        timings["allgather_s"] = timings.get("allgather_s", 0.0) + (time.perf_counter() - t_shard)
        timings["calls"] = timings.get("calls", 0) + 1
    mem = unpack_memory(gathered, world, engine.L, engine.N, engine.d, q.shape[1], engine.dm, engine.H if with_scores else 0)
    return ctx, mem
