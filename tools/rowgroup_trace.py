#!/usr/bin/env python
"""Phase times of the row-group builder (s_memtime stamps per window): sample search | table -> LDS | masks | sort |
perm + gmask | emit.   B=16 python tools/rowgroup_trace.py"""
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
buf = torch.zeros((8192, 8), dtype=torch.int64, device="cuda")
ctx.lib.egonn_debug_set_trace(buf.data_ptr())
for kind, lvl in [(0, 1), (1, 1), (0, 4)]:
    buf.zero_()
    ctx.map_groups(kind, lvl)                    # builds this map's tables (one launch)
    torch.cuda.synchronize()
    t = buf.cpu().numpy()
    t = t[t[:, 6] > 0]
    names = ["search", "load", "masks", "sort", "perm", "emit"]
    d = np.stack([t[:, i + 1] - t[:, i] for i in range(6)], 1).astype(np.float64)
    full = t[:, 7] >> 8
    print(f"map kind {kind} L{lvl}: {len(t)} windows (K = {int(t[0, 7] & 255)}, median rows {int(np.median(full))}); cycles per phase (median / p90):")
    print("   " + "  ".join(f"{n} {np.median(d[:, i]):.0f}/{np.percentile(d[:, i], 90):.0f}" for i, n in enumerate(names)),
          f"  total {np.median(d.sum(1)):.0f}")
ctx.lib.egonn_debug_set_trace(None)
