import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from infinite_video_amd import synth
from infinite_video_amd.engine import LTMEngine
from infinite_video_amd.video_memory import consolidate_video
T, P, D, N, H, DH, Q, L, TAU = 256, 32, 768, 256, 12, 64, 32, 2, 0.75
dev = torch.device("cuda:0")
mb = int(sys.argv[1])
eng = LTMEngine(N, H, DH, D, P, tau=TAU, sticky=True, n_layers=L, max_q=Q, device=dev, max_batch_chunks=mb)
projs = [tuple(torch.from_numpy(a).to(dev) for a in synth.layer_projections(l, D, H * DH)) for l in range(L)]
q = torch.from_numpy(np.stack([synth.layer_query(l, Q, H * DH) for l in range(L)])).to(dev)
u = torch.from_numpy(synth.gibbs_uniforms(256, L)).to(dev)
k = torch.empty(256, T * P, D, device=dev).normal_()
ts = []
for p in range(9):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx, mem = consolidate_video(eng, k, q, projs, u)
    torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print("max_batch", mb, "ramp", os.environ.get("INFV_SUB_RAMP"), "median ms", sorted(ts[2:])[3], "min", min(ts[2:]), "checksum", float(ctx.double().sum()))
