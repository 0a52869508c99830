#!/usr/bin/env python
"""Dump the 27-bit neighbour-presence mask of every row (in Z-order) of levels 1-4 of the benchmark batch, plus the
per-sample row offsets -> gpurun_out/masks.npz (input of tools/grouping_study.py, which runs on the CPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g; g.build()
from egonn_amd import _lib
from egonn_amd.synth import lidar_scan
B = int(os.environ.get("B", 16))
scans = [lidar_scan(1000 + i, 50000) for i in range(B)]
off = [0]
for s in scans: off.append(off[-1] + len(s))
pts = torch.from_numpy(np.concatenate(scans)).cuda()
ctx = _lib.Context(coord_bits=12)
ctx.voxelize(pts, off, 0, [0.1])
out = {}
for lvl in (1, 2, 3, 4, 5):
    gm, sn = ctx.rowgroup_tables(0, lvl)
    ng = len(gm)
    sn = sn.cpu().numpy()                                    # [g][27][16]
    n = ctx.level_count(lvl)
    # perm is not exported: rebuild masks per slot, then order does not matter for the study except windows; export per-slot masks
    bits = (sn >= 0).astype(np.uint32)                        # [g][27][16]
    m = (bits * (1 << np.arange(27, dtype=np.uint32))[None, :, None]).sum(axis=1).astype(np.uint32)   # [g][16]
    out[f"slotmask_{lvl}"] = m
    _, first = ctx.map_groups(0, lvl)
    out[f"first_{lvl}"] = np.array(first)
    out[f"n_{lvl}"] = n
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/masks.npz", **out)
print("ok", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})
