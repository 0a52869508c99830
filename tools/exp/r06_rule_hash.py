#!/usr/bin/env python
"""sha1 of stand-alone sparse-conv outputs on the benchmark maps under the rule of the environment (EGONN_KSPLIT_KW ...):
two runs with different rules must print the same digests when the rule only changes the schedule."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from egonn_amd import _lib
from egonn_amd.synth import lidar_scan
B = int(os.environ.get("B", 4))
scans = [lidar_scan(1000 + i, 50000) for i in range(B)]
off = [0]
for s in scans: off.append(off[-1] + len(s))
pts = torch.from_numpy(np.concatenate(scans)).cuda()
ctx = _lib.Context(coord_bits=12); ctx.voxelize(pts, off, 0, [0.1])
for (kind, lvl, ci, co) in [(0, 4, 64, 128), (0, 4, 128, 128), (1, 5, 128, 128), (0, 5, 128, 128), (2, 4, 128, 128), (0, 3, 64, 64)]:
    lin = lvl if kind == 0 else (lvl - 1 if kind == 1 else lvl + 1)
    K = 27 if kind == 0 else 8
    torch.manual_seed(lvl * 100 + ci)
    x = torch.randn(ctx.level_count(lin), ci, device="cuda")
    w = torch.randn(K, ci, co, device="cuda") * (1.0 / np.sqrt(ci * (9 if K == 27 else 2)))
    sc = torch.rand(co, device="cuda") + 0.5; sh = torch.randn(co, device="cuda")
    got = ctx.sparse_conv(kind, lvl, x, w, sc, sh, True)
    print(kind, lvl, ci, co, hashlib.sha1(got.cpu().numpy().tobytes()).hexdigest()[:16], bool(torch.isfinite(got).all()))
