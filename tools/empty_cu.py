#!/usr/bin/env python3
"""Where the empty CU-time of a headline pass goes (review item 7 of round 5).  Input: a stamp file written by tools/residency.py
(experiments build, INFV_WG_STAMPS=1): [start, end, (xcc << 32) | hw_id, kind] per workgroup, launches appended in issue order.
For every interval in which a CU hosts no workgroup of the five pipeline kernels: which kernel's workgroup lands next, how long
the CU stayed empty; per pooling launch: when each XCD finished its eighth of the grid (blocks are dealt round-robin over the XCDs
at launch, so a launch ends with its slowest XCD); per XCD: time-weighted pooling residents.
usage: python tools/empty_cu.py gpurun_out/wg_stamps_<tag>.npy [pool_grid]"""
import sys
import numpy as np

st = np.load(sys.argv[1])
pool_grid = int(sys.argv[2]) if len(sys.argv) > 2 else 2688
idx = np.arange(len(st))
ok = (st[:, 1] > 0) & (st[:, 0] > 0)
st, idx = st[ok], idx[ok]
names = {1: "pool", 2: "gemm", 3: "uc", 4: "roleS", 5: "alpha"}
t0 = st[:, 0].min()
start, end = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0
hw, xcc, kind = st[:, 2] & 0xffffffff, st[:, 2] >> 32, st[:, 3]
cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | ((hw >> 8) & 15)
cus = np.unique(cu)
span = end.max()
print(f"records {len(st)}  span {span/1e3:.2f} ms  CUs {len(cus)}")

# ---- empty intervals per CU: who lands next ----
tot_empty = 0.0
by_next = {k: [0.0, 0] for k in list(names) + [0]}
hist_edges = [0, 2, 5, 10, 20, 50, 100, 200, 1e9]
hist = {k: np.zeros(len(hist_edges) - 1) for k in by_next}
prev_kind_time = {k: 0.0 for k in names}            # empty time by the kind of the LAST workgroup that left the CU
for c in cus:
    m = cu == c
    o = np.argsort(start[m])
    s, e, kd = start[m][o], end[m][o], kind[m][o]
    cur_e, last_kind = 0.0, 0
    # running max of ends with the kind of the workgroup that ended last
    for a, b, k in zip(s, e, kd):
        if a > cur_e:
            gap = a - cur_e
            tot_empty += gap
            by_next[k][0] += gap; by_next[k][1] += 1
            hist[k][np.searchsorted(hist_edges, gap, "right") - 1] += gap
            if last_kind: prev_kind_time[last_kind] += gap
        if b > cur_e:
            cur_e, last_kind = b, k
    if span > cur_e:
        tot_empty += span - cur_e
        by_next[0][0] += span - cur_e; by_next[0][1] += 1
cu_time = len(cus) * span
print(f"empty CU-time {tot_empty/cu_time:.3f} of the pass ({tot_empty/1e3:.0f} CU.ms)")
print("  by the kernel whose workgroup lands NEXT on the empty CU: share of the pass's CU-time, intervals, mean us; by length of the interval (us):")
print("          share  count    mean   " + "".join(f"{'<%g' % h:>8s}" for h in hist_edges[1:-1]) + "    more")
for k, (t, n) in by_next.items():
    if n == 0: continue
    nm = names.get(k, "(end)")
    print(f"  {nm:6s} {t/cu_time:6.3f} {n:6d} {t/max(n,1):7.1f}   " + "".join(f"{v/cu_time:8.3f}" for v in hist[k]))
print("  by the kernel whose workgroup LEFT last: " + "  ".join(f"{names[k]} {v/cu_time:.3f}" for k, v in prev_kind_time.items()))

# ---- pooling launches: per-XCD completion ----
pm = kind == 1
pidx = idx[pm]
if pm.any():
    # launches = runs of consecutive records of kind 1 (records are appended per launch, in issue order); two pooling launches
    # issued back to back (the call's first two sub-batches) form one run: split at the grid size
    order = np.argsort(pidx)
    pi, ps, pe, px = pidx[order], start[pm][order], end[pm][order], xcc[pm][order]
    brk = np.flatnonzero(np.diff(pi) > 1) + 1
    launches = []
    for g in np.split(np.arange(len(pi)), brk):
        if len(g) > pool_grid and len(g) % pool_grid == 0:
            for j in range(0, len(g), pool_grid): launches.append(g[j:j + pool_grid])
        else:
            launches.append(g)
    rows = []
    for g in launches:
        if len(g) < pool_grid: continue
        s0 = ps[g].min()
        fin = np.array([pe[g][px[g] == x].max() - s0 for x in range(8)])
        first = np.array([ps[g][px[g] == x].min() - s0 for x in range(8)])
        life = np.array([np.median(pe[g][px[g] == x] - ps[g][px[g] == x]) for x in range(8)])
        rows.append((fin, first, life, pe[g].max() - s0))
    fin = np.array([r[0] for r in rows]); life = np.array([r[2] for r in rows]); dur = np.array([r[3] for r in rows])
    print(f"pooling launches of {pool_grid} workgroups: {len(rows)}; first workgroup start -> last end: median {np.median(dur):.1f} us")
    print("  per-XCD completion of its eighth, relative to the launch's length (median over launches; 1.00 = the slowest XCD):")
    print("   " + " ".join(f"{v:6.2f}" for v in np.median(fin / dur[:, None], axis=0)))
    print(f"  fastest XCD finishes at {np.median(fin.min(axis=1) / dur):.2f} of the launch; mean over XCDs {np.median(fin.mean(axis=1) / dur):.2f}"
          f"  -> {1 - np.median(fin.mean(axis=1) / dur):.2f} of the pooling stream's seat-time is spent waiting for the slowest XCD")
    print("  median workgroup lifetime per XCD (us): " + " ".join(f"{v:6.1f}" for v in np.median(life, axis=0)))
    # which XCD is the slowest, how often
    slow = fin.argmax(axis=1)
    print("  launches in which XCD x was the slowest: " + " ".join(f"{(slow == x).sum():4d}" for x in range(8)))
# ---- per XCD residents ----
print("time-weighted residents per XCD (32 CUs each):")
print("          " + "".join(f"{'xcd%d' % x:>7s}" for x in range(8)))
for kd, nm in names.items():
    m = kind == kd
    if not m.any(): continue
    print(f"  {nm:6s}  " + "".join(f"{(end[m & (xcc == x)] - start[m & (xcc == x)]).sum()/span:7.1f}" for x in range(8)))
