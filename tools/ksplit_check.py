#!/usr/bin/env python
"""Correctness of the offset-split launches against the plain one-thread-per-output kernel and the unsplit lock-step kernel
(stand-alone operator calls on the benchmark maps).  Run with EGONN_KSPLIT / EGONN_KSPLIT8 / EGONN_SPLIT_MAX_LEVEL set."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g; g.build()
from egonn_amd import _lib
from egonn_amd.synth import lidar_scan
B = int(os.environ.get("B", 4))
scans = [lidar_scan(1000 + i, 50000) for i in range(B)]
off = [0]
for s in scans: off.append(off[-1] + len(s))
pts = torch.from_numpy(np.concatenate(scans)).cuda()
ctx = _lib.Context(coord_bits=12); ctx.voxelize(pts, off, 0, [0.1])
ref = _lib.Context(coord_bits=12); ref.voxelize(pts, off, 0, [0.1]); ref.set_naive_conv(True)
cfgs = [(0, 3, 64, 64), (1, 4, 64, 64), (0, 4, 64, 128), (0, 4, 128, 128), (1, 5, 128, 128), (0, 5, 128, 128), (0, 6, 128, 128),
        (0, 7, 128, 128), (1, 7, 128, 128), (2, 6, 128, 128), (2, 5, 128, 128), (2, 3, 64, 64)]
worst = 0.0
for (kind, lvl, ci, co) in cfgs:
    lin = lvl if kind == 0 else (lvl - 1 if kind == 1 else lvl + 1)
    K = 27 if kind == 0 else 8
    n_in = ctx.level_count(lin)
    torch.manual_seed(lvl * 100 + ci)
    x = torch.randn(n_in, ci, device="cuda")
    w = torch.randn(K, ci, co, device="cuda") * (1.0 / np.sqrt(ci * (9 if K == 27 else 2)))
    sc = torch.rand(co, device="cuda") + 0.5; sh = torch.randn(co, device="cuda")
    want = ref.sparse_conv(kind, lvl, x, w, sc, sh, True)
    got = ctx.sparse_conv(kind, lvl, x, w, sc, sh, True)
    got2 = ctx.sparse_conv(kind, lvl, x, w, sc, sh, True)
    err = float(((got - want).abs().max() / (want.abs().max() + 1e-9)).item())
    worst = max(worst, err)
    print(f"kind {kind} L{lvl} {ci}->{co}: rel err {err:.2e}  rerun bitwise {bool((got == got2).all())}  finite {bool(torch.isfinite(got).all())}", flush=True)
assert worst < 3e-6, worst
print("ksplit check ok", worst)
