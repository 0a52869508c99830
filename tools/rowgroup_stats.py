#!/usr/bin/env python
"""How well do the row groups pack?  For every conv map of the benchmark batch: groups, (group, offset) items the MFMA
kernel walks, pairs, MFMA padding = 16 * items / pairs, and the histogram of real rows per item (1..16).
    B=16 python tools/rowgroup_stats.py"""
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
out = []
for kind, lvl in [(0, 1), (0, 2), (0, 3), (0, 4), (0, 5), (0, 6), (0, 7), (1, 1), (1, 2), (1, 3), (2, 3)]:
    gm, sn = ctx.rowgroup_tables(kind, lvl)
    gm = gm.cpu().numpy().astype(np.uint32)
    sn = sn.cpu().numpy()
    rows_per_item = (sn >= 0).sum(axis=2)                 # [groups][K]
    items = int((rows_per_item > 0).sum())
    pairs = int(rows_per_item.sum())
    popc = np.array([bin(int(m) & 0x7FFFFFF).count("1") for m in gm])
    assert popc.sum() == items, (popc.sum(), items)
    hist = np.bincount(rows_per_item[rows_per_item > 0], minlength=17)[1:]
    real_rows = ctx.level_count(lvl)
    rec = dict(kind=kind, level=lvl, rows=real_rows, groups=len(gm), items=items, pairs=pairs, mfma_pad=16 * items / pairs,
               items_per_group=items / len(gm), hist=hist.tolist(),
               items_le2=float(hist[:2].sum() / items), pairs_in_le2=float((hist[:2] * np.arange(1, 3)).sum() / pairs),
               items_le4=float(hist[:4].sum() / items), pairs_in_le4=float((hist[:4] * np.arange(1, 5)).sum() / pairs))
    out.append(rec)
    print(f"kind {kind} L{lvl}: rows {real_rows} groups {len(gm)} items {items} ({items / len(gm):.1f}/group) pairs {pairs} pad {rec['mfma_pad']:.2f}  "
          f"items with <=2 rows {rec['items_le2']:.2f} (hold {rec['pairs_in_le2']:.3f} of pairs), <=4 rows {rec['items_le4']:.2f} ({rec['pairs_in_le4']:.3f})")
    print("    rows/item hist 1..16:", hist.tolist(), flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
