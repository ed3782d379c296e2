#!/usr/bin/env python3
"""Fill and drain of one 256-chunk shard call: first start / last end of every launch of the pipeline kernels (from residency
stamps, experiments build with INFV_WG_STAMPS=1).  usage: python tools/shard_timeline.py gpurun_out/wg_stamps_<tag>.npy <pool_grid>"""
import sys
import numpy as np
st = np.load(sys.argv[1]); pool_grid = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
t0 = st[st[:, 0] > 0, 0].min()
kind = st[:, 3]
names = {1: "pool", 2: "gemm", 3: "uc", 4: "roleS", 5: "alpha"}
out = []
i, n = 0, len(st)
while i < n:
    k = int(kind[i]); j = i
    while j < n and kind[j] == k: j += 1
    blk = st[i:j]
    sz = {1: pool_grid, 4: 48, 3: 144}.get(k)
    parts = [blk] if (sz is None or len(blk) <= sz) else [blk[a:a + sz] for a in range(0, len(blk), sz)]
    for p in parts:
        ok = (p[:, 0] > 0) & (p[:, 1] > 0)
        if ok.any(): out.append(((p[ok, 0].min() - t0) / 100., (p[ok, 1].max() - t0) / 100., names[k], int(ok.sum())))
    i = j
out.sort()
for s, e, nm, cnt in out: print(f"{s:8.0f} - {e:8.0f} ({e-s:5.0f} us)  {nm:6s} {cnt:5d} workgroups")
print("span", max(e for _, e, _, _ in out))
