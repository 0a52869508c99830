#!/usr/bin/env python
"""CPU study for the window-resident gather kernel: how many DISTINCT input rows does a window of W consecutive
Z-order rows of a level reference through its 27-offset map?  (benchmark clouds, Cartesian 0.1 m)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from egonn_amd.synth import lidar_scan

def morton(c):
    c = c.astype(np.uint64)
    def spread(v):
        r = np.zeros_like(v)
        for b in range(16):
            r |= ((v >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b)
        return r
    return spread(c[:, 0]) | (spread(c[:, 1]) << np.uint64(1)) | (spread(c[:, 2]) << np.uint64(2))

def level_nbr(c):
    """c: (n,3) int coords (unit stride at this level), returns rows in Z-order and nbr (n,27)"""
    c = c + 2048
    order = np.argsort(morton(c), kind="stable")
    c = c[order]
    lin = (c[:, 0].astype(np.int64) << 26) | (c[:, 1].astype(np.int64) << 13) | c[:, 2].astype(np.int64)
    so = np.argsort(lin); ls = lin[so]
    nbr = np.full((len(c), 27), -1, np.int64)
    k = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                q = ((c[:, 0] + dx).astype(np.int64) << 26) | ((c[:, 1] + dy).astype(np.int64) << 13) | (c[:, 2] + dz).astype(np.int64)
                p = np.searchsorted(ls, q); p[p >= len(ls)] = len(ls) - 1
                hit = ls[p] == q
                nbr[hit, k] = so[p[hit]]
                k += 1
    return nbr

res = {}
for seed in range(1000, 1004):
    pts = lidar_scan(seed, 50000)
    c0 = np.unique(np.floor(pts / np.float32(0.1)).astype(np.int64), axis=0)
    for lvl in (1, 2, 3, 4):
        c = np.unique(c0 >> lvl, axis=0)
        nbr = level_nbr(c)
        n = len(c)
        for W in (64, 128, 256, 512):
            d, h, pr = [], [], []
            for r0 in range(0, n, W):
                t = nbr[r0:r0 + W]
                rows = len(t)
                v = t[t >= 0]
                u = np.unique(v)
                halo = ((u < r0) | (u >= r0 + rows)).sum()
                d.append(rows + halo); h.append(halo); pr.append(len(v))
            res.setdefault((lvl, W), []).append((np.array(d), np.array(h), np.array(pr), n))
for (lvl, W), lst in sorted(res.items()):
    d = np.concatenate([x[0] for x in lst]); h = np.concatenate([x[1] for x in lst]); pr = np.concatenate([x[2] for x in lst])
    n = sum(x[3] for x in lst)
    full = d[h + W == d] if False else d
    print(f"L{lvl} W={W:4d}: windows {len(d):5d} rows/scan {n // len(lst):6d} pairs/row {pr.sum() / n:5.2f}  distinct/window mean {d.mean():6.1f} "
          f"p50 {np.percentile(d, 50):5.0f} p90 {np.percentile(d, 90):5.0f} p99 {np.percentile(d, 99):5.0f} max {d.max():5d}  "
          f"halo mean {h.mean():6.1f} max {h.max():4d}  staged/own {d.sum() / n:4.2f}")
