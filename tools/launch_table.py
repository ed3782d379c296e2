#!/usr/bin/env python3
"""Per-launch timeline (first workgroup start .. last workgroup end, us) of the five pipeline kernels from a residency stamp file
written by tools/residency.py.  usage: python tools/launch_table.py gpurun_out/wg_stamps_<tag>.npy [first_sub_batch] [count]"""
import sys
import numpy as np
st = np.load(sys.argv[1])
b0 = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 7
t0 = st[st[:, 0] > 0, 0].min()
kind = st[:, 3]
out = {k: [] for k in range(1, 6)}
i, n = 0, len(st)
while i < n:                                   # records are appended per launch: runs of one kind, split by the known grid sizes
    k = kind[i]; j = i
    while j < n and kind[j] == k: j += 1
    blk = st[i:j]
    sz = {1: 2688, 4: 48, 3: 144}.get(int(k))
    parts = [blk] if (sz is None or len(blk) <= sz) else [blk[a:a + sz] for a in range(0, len(blk), sz)]
    for p in parts:
        ok = (p[:, 0] > 0) & (p[:, 1] > 0)
        if ok.any(): out[int(k)].append(((p[ok, 0].min() - t0) / 100., (p[ok, 1].max() - t0) / 100.))
    i = j
names = {1: "pool", 2: "gemm", 4: "roleS", 5: "alpha", 3: "uc"}
print("launches:", {names[k]: len(v) for k, v in out.items()}, "(uc has one launch more: the first chunk)")
for b in range(b0, b0 + cnt):
    row = f" b={b}: "
    for k in [1, 2, 4, 5, 3]:
        bb = b + 1 if k == 3 else b
        if bb < len(out[k]): s, e = out[k][bb]; row += f"{names[k]} {s:8.0f}-{e:8.0f} ({e-s:4.0f}) | "
    print(row)
