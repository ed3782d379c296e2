#!/usr/bin/env python3
"""Who shares a CU with whom during one headline pass (experiments build, INFV_WG_STAMPS=1): every workgroup of the pooling,
projection-GEMM, UC, role-S and alpha-rows kernels records start, end and CU.  Prints per kernel: workgroups, lifetime,
time-weighted residents, and for each ordered pair (A, B) the share of A's workgroup-time during which a B workgroup was
resident on the same CU.
usage (GPU box): INFV_LTM_LIBRARY=exp INFV_WG_STAMPS=1 python tools/residency.py [tag] [chunks]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from infinite_video_amd import _lib, synth
from infinite_video_amd.engine import LTMEngine
from infinite_video_amd.video_memory import consolidate_video

T, P, D, N, H, DH, Q, L, TAU = 256, 32, 768, 256, 12, 64, 32, 2, 0.75
tag = sys.argv[1] if len(sys.argv) > 1 else "run"
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
dev = torch.device("cuda:0")
eng = LTMEngine(N, H, DH, D, P, tau=TAU, sticky=True, n_layers=L, max_q=Q, device=dev, max_batch_chunks=42)
projs = [tuple(torch.from_numpy(a).to(dev) for a in synth.layer_projections(l, D, H * DH)) for l in range(L)]
q = torch.from_numpy(np.stack([synth.layer_query(l, Q, H * DH) for l in range(L)])).to(dev)
u = torch.from_numpy(synth.gibbs_uniforms(chunks, L)).to(dev)
k = torch.empty(chunks, T * P, D, device=dev)
gen = torch.Generator(device=dev).manual_seed(1)
for i in range(0, chunks, 64):
    k[i:i + 64].normal_(generator=gen)
lib = _lib.load()
fn = lib.infv_exp_wg_stamps
fn.restype = C.c_long
fn.argtypes = [C.c_void_p, C.c_long]
cap = 1 << 19
buf = np.zeros((cap, 4), np.int64)
for p in range(3):
    consolidate_video(eng, k, q, projs, u)
    torch.cuda.synchronize()
    n = fn(buf.ctypes.data, cap)
st = buf[:n].copy()
np.save(os.path.join(ROOT, "gpurun_out", f"wg_stamps_{tag}.npy"), st)
st = st[(st[:, 1] > 0) & (st[:, 0] > 0)]
names = {1: "pool", 2: "gemm", 3: "uc", 4: "roleS", 5: "alpha"}
t0 = st[:, 0].min()
start, end = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0          # us
hw, xcc, kind = st[:, 2] & 0xffffffff, st[:, 2] >> 32, st[:, 3]
cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | ((hw >> 8) & 15)
cus = np.unique(cu)
span = end.max()
print(f"[{tag}] records {len(st)}, span {span/1e3:.2f} ms, distinct CUs {len(cus)}")
for kd, nm in names.items():
    m = kind == kd
    if not m.any():
        continue
    life = end[m] - start[m]
    print(f"  {nm:6s} workgroups {m.sum():7d}  lifetime us median {np.median(life):7.1f} p10 {np.percentile(life,10):7.1f} p90 {np.percentile(life,90):7.1f}"
          f"  residents (time-weighted over the span) {life.sum()/span:6.1f}")
# overlap matrix: for A's workgroup-time, the share with a B resident on the same CU
present = [kd for kd in names if (kind == kd).any()]
busy = {}                                                # kind -> per-CU list of merged [s, e] intervals
for kd in present:
    per = {}
    for c in cus:
        m = (kind == kd) & (cu == c)
        if not m.any():
            per[c] = (np.zeros(0), np.zeros(0)); continue
        o = np.argsort(start[m]); s, e = start[m][o], end[m][o]
        ms, me = [s[0]], [e[0]]
        for a, b in zip(s[1:], e[1:]):
            if a <= me[-1]: me[-1] = max(me[-1], b)
            else: ms.append(a); me.append(b)
        per[c] = (np.array(ms), np.array(me))
    busy[kd] = per
def covered(s, e, ms, me):
    """length of [s, e] covered by the merged intervals (ms, me)"""
    if len(ms) == 0: return 0.0
    i0 = np.searchsorted(me, s, "right"); i1 = np.searchsorted(ms, e, "left")
    if i1 <= i0: return 0.0
    return float((np.minimum(me[i0:i1], e) - np.maximum(ms[i0:i1], s)).clip(min=0).sum())
print("  share of A's workgroup-time with a B workgroup on the same CU (rows A, columns B):")
print("          " + "".join(f"{names[b]:>8s}" for b in present) + "   none-of-the-others")
for a in present:
    m = kind == a
    tot = (end[m] - start[m]).sum()
    row = []
    for b in present:
        if b == a: row.append(float("nan")); continue
        acc = 0.0
        for c in cus:
            mm = m & (cu == c)
            ms, me = busy[b][c]
            for s, e in zip(start[mm], end[mm]): acc += covered(s, e, ms, me)
        row.append(acc / tot)
    print(f"  {names[a]:>8s}" + "".join(f"{v:8.2f}" for v in row))
# CU-time with nothing resident at all / with no pooling workgroup
allms = {}
idle = 0.0; nopool = 0.0
for c in cus:
    m = cu == c
    o = np.argsort(start[m]); s, e = start[m][o], end[m][o]
    cur_e = 0.0; cov = 0.0
    for a, b in zip(s, e):
        if b <= cur_e: continue
        cov += b - max(a, cur_e); cur_e = b
    idle += span - cov
    ms, me = busy[1][c] if 1 in busy else (np.zeros(0), np.zeros(0))
    nopool += span - float((me - ms).sum())
print(f"  CU-time with no workgroup of these kernels resident: {idle/(len(cus)*span):.2f};  with no pooling workgroup resident: {nopool/(len(cus)*span):.2f}")
