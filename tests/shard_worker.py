"""Child program of tests/test_sharding_gpu.py: ONE rank of a world-size-2 run on ONE GPU.

Fresh process (started with subprocess by the test; nothing here is imported into the pytest process), backend
``gloo`` (two RCCL ranks cannot share a device), a real ``LTMEngine`` on ``cuda:0``.  Runs
``video_memory.consolidate_video`` on this rank's block of the video -- independent documents (default) or the exact
hand-off mode -- and writes its outputs to ``<out>/rank<r>.npz``.

usage: python -m tests.shard_worker <rank> <world> <port> <mode: shard|handoff> <n_chunks> <out_dir>
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    mode, n_chunks, out = sys.argv[4], int(sys.argv[5]), sys.argv[6]
    from infinite_video_amd.video_memory import consolidate_video, shard_range
    from tests.test_timed_path_gpu import L, Q, _engine, _video
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        k, q, projs, u, _, _ = _video(dev, n_chunks)               # every rank generates the same video; it keeps its block
        a, b = shard_range(n_chunks, world, rank)
        eng = _engine(dev, max_batch_chunks=42)
        ctx, mem = consolidate_video(eng, k[a:b].contiguous(), q, projs, u[a:b].contiguous(), handoff=(mode == "handoff"))
        torch.cuda.synchronize()
        np.savez(os.path.join(out, f"rank{rank}.npz"), ctx=ctx.cpu().numpy(), B=mem.B.cpu().numpy(),
                 bin_mass=mem.bin_mass.cpu().numpy(), ctx_sum=mem.ctx_sum.cpu().numpy(), count=mem.count.cpu().numpy(),
                 bins=np.stack([eng.last_draw(l)[0] for l in range(L)]))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
