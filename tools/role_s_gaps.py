#!/usr/bin/env python3
"""Gaps between consecutive role-S launches of a pass, from a residency stamp file (tools/residency.py): last workgroup end of launch b
-> first workgroup start of launch b + 1, the start skew of a launch's workgroups, the launches' own length.
usage: python tools/role_s_gaps.py gpurun_out/wg_stamps_<tag>.npy"""
import sys
import numpy as np
st = np.load(sys.argv[1])
ok = (st[:, 0] > 0) & (st[:, 1] > 0)
kind = st[:, 3]
idx = np.flatnonzero((kind == 4) & ok)
runs = np.split(idx, np.flatnonzero(np.diff(idx) > 1) + 1)
t0 = st[ok, 0].min()
rows = np.array([((st[r, 0].min() - t0) / 100, (st[r, 0].max() - t0) / 100, (st[r, 1].min() - t0) / 100, (st[r, 1].max() - t0) / 100) for r in runs])
gaps, skew, dur = rows[1:, 0] - rows[:-1, 3], rows[:, 1] - rows[:, 0], rows[:, 3] - rows[:, 0]
print(f"role-S launches {len(rows)}: length median {np.median(dur):.1f} us; start skew of a launch's workgroups median {np.median(skew):.1f} (p90 {np.percentile(skew, 90):.1f})")
print(f"gap between launches: median {np.median(gaps):.1f} us, p10 {np.percentile(gaps, 10):.1f}, p90 {np.percentile(gaps, 90):.1f}; sum {gaps.sum() / 1e3:.2f} ms of a {(rows[-1, 3] - rows[0, 0]) / 1e3:.2f} ms span (launches {dur.sum() / 1e3:.2f} ms)")
