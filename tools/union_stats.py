#!/usr/bin/env python
"""Union-walk inflation of the row groups: steps a wave / workgroup that owns n consecutive groups walks (popcount of the OR
of their masks) x n against the sum of the groups' own popcounts.   B=16 python tools/union_stats.py"""
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
POP = np.array([bin(i).count("1") for i in range(1 << 16)], dtype=np.int64)
popc = lambda x: POP[x & 0xFFFF] + POP[(x >> 16) & 0x7FF]
for kind, lvl in [(0, 1), (0, 2), (0, 3), (0, 4), (0, 5), (0, 6), (0, 7), (1, 1), (1, 2), (1, 3), (1, 5)]:
    gm, sn = ctx.rowgroup_tables(kind, lvl)
    gm = gm.cpu().numpy().astype(np.uint32)
    live = (gm >> 31) != 0
    own = popc(gm).sum()
    snc = sn.cpu().numpy()
    K = snc.shape[1]
    bits = ((gm[:, None] >> np.arange(K)[None, :]) & 1).astype(bool)          # [groups, K]
    pairs = int(((snc >= 0) & bits[:, :, None]).sum())
    rows = int((snc >= 0).any(axis=1).sum()) if K == 27 else 0
    line = f"kind {kind} L{lvl}: groups {len(gm)} live {int(live.sum())} items {own} ({own / live.sum():.2f}/live group) pairs {pairs} fill {pairs / (16.0 * own):.3f}"
    # compaction alternative: blocks of R consecutive rows (R/16 groups), per offset ceil(cnt/16) tiles instead of one per group with the bit
    for R in (64, 128, 256):
        g = R // 16
        m = (snc[: len(snc) // g * g] >= 0) & bits[: len(snc) // g * g, :, None]
        cnt = m.reshape(-1, g, K, 16).sum(axis=(1, 3))                          # [blocks, K]
        tiles = int(np.ceil(cnt / 16.0).sum())
        line += f" | R={R}: tiles {tiles} ({tiles / own:.2f} of now)"
    for n in (2, 4, 8, 16):
        m = gm[: len(gm) // n * n].reshape(-1, n)
        u = np.bitwise_or.reduce(m, axis=1)
        anylive = (u >> 31) != 0
        steps = popc(u)[anylive].sum()
        line += f" | n={n}: steps {steps} x{n} = {steps * n / own:.2f}"
    print(line, flush=True)
