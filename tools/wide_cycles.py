#!/usr/bin/env python3
"""Main-loop cycles of split_gemm_wide_kernel in situ (experiments build, INFV_WG_STAMPS=1): every workgroup stamps the 100 MHz
clock and the shader clock around its k-loop.  Prints, per contraction (24 k-tiles: scores, 84-86: read-out), the shader cycles per
32-deep k-tile (144 MFMAs per SIMD = 4608 cycles at the matrix pipe's rate) and the clock the chip held inside the loop.
usage (GPU box): INFV_LTM_LIBRARY=exp INFV_WG_STAMPS=1 python tools/wide_cycles.py [chunks]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from infinite_video_amd import _lib, synth
from infinite_video_amd.video_qformer import InfVideoEncoder

chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 56
dev = torch.device("cuda:0")
m = InfVideoEncoder(num_basis=256, tau=0.75, alpha=0.9, sticky=True)
m.load_reference_state_dict(synth.video_qformer_weights())
m = m.to(dev)
frames = torch.randn(chunks, 256 * 32, 768, device=dev)
u = torch.from_numpy(synth.gibbs_uniforms(chunks, 2)).to(dev)
lib = _lib.load()
fn = lib.infv_exp_wg_stamps
fn.restype = C.c_long
fn.argtypes = [C.c_void_p, C.c_long]
cap = 1 << 19
buf = np.zeros((cap, 4), np.int64)
for p in range(3):
    m.encode_frames_batch(frames, new_video=True, u=u)
    torch.cuda.synchronize()
    n = fn(buf.ctypes.data, cap)
st = buf[:n]
cyc = st[:, 2] >> 36
st = st[(cyc > 0) & (st[:, 1] > st[:, 0])]
cyc = (st[:, 2] >> 36).astype(np.float64)
dt_us = (st[:, 1] - st[:, 0]) / 100.0
print(f"{len(st)} workgroups of split_gemm_wide_kernel stamped (last of 3 passes, {chunks} chunks)")
for name, lo, hi, tiles in (("scores   (K = 768, 24 k-tiles)", 0, 100 * 2304, 24), ("read-out (K = 8192 / 3, 84-86 k-tiles)", 100 * 2304, 1 << 40, 86)):
    sel = (cyc >= lo) & (cyc < hi)
    if not sel.any():
        continue
    c, d = cyc[sel], dt_us[sel]
    print(f"{name}: {sel.sum()} workgroups, loop {np.median(d):.1f} us (p10 {np.percentile(d, 10):.1f}, p90 {np.percentile(d, 90):.1f}), "
          f"{np.median(c) / tiles:.0f} cycles per k-tile (4608 = matrix pipe busy all the time: {100 * 4608 * tiles / np.median(c):.0f} %), "
          f"clock in the loop {np.median(c / d) / 1e3:.2f} GHz")
del m          # (before interpreter teardown)
