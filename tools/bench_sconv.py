#!/usr/bin/env python
"""Micro-benchmark of the sparse-conv kernel variants on the benchmark workload (tuning aid)."""
import sys, os, json
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
lib = _lib.load()
cfgs = [(1, 32, 32, 3)] if os.environ.get("ONLY32") else [(1, 32, 32, 3), (2, 32, 64, 3), (2, 64, 64, 3), (3, 64, 64, 3), (4, 128, 128, 3), (6, 128, 128, 3), (1, 32, 32, 2)]
res = {}
for var in map(int, os.environ.get("VARS", "0,7,16,23").split(",")):
    lib.egonn_debug_set_naive_conv(0x100 | ((var & 7) << 4) | (((var >> 4) & 3) << 12) | ((var >> 8) << 16))   # var bits 8-9: skip W / skip A loads
    for (lvl, ci, co, ks) in cfgs:
        lin = lvl if ks == 3 else lvl - 1
        x = torch.randn(ctx.level_count(lin), ci, device="cuda")
        w = torch.randn(27 if ks == 3 else 8, ci, co, device="cuda") * 0.05
        for _ in range(3): ctx.conv(lin, lvl, ks, x, w)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ctx.conv(lin, lvl, ks, x, w)
        e1.record(); torch.cuda.synchronize()
        res[(var, lvl, ci, co, ks)] = e0.elapsed_time(e1) / 20 * 1e3
        print(f"var {var} L{lvl} {ci}->{co} k{ks}: {res[(var, lvl, ci, co, ks)]:8.1f} us", flush=True)
