"""probe: rows of a scan alone vs inside a batch under the in-workgroup offset parts (KW)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import __graft_entry__ as g; g.build()
from egonn_amd import _lib
from egonn_amd.synth import lidar_scan
def plan(seeds):
    scans = [lidar_scan(s, 30000) for s in seeds]
    off = [0]
    for s in scans: off.append(off[-1] + len(s))
    ctx = _lib.Context(coord_bits=12); ctx.voxelize(torch.from_numpy(np.concatenate(scans)).cuda(), off, 0, [0.1]); return ctx
seeds = [410, 411, 412, 413]
batch = plan(seeds)
for which in (0, 1, 2, 3):
    alone = plan([seeds[which]])
    for (kind, lvl, ci, co) in [(0, 4, 128, 128), (0, 5, 128, 128), (0, 3, 64, 64)]:
        for kw in (0, 2, 3, 4):
            for c in (batch, alone): c.set_ksplit(0, lvl, kparts=1, kw=kw, col_parts=0)
            gen = torch.Generator(device="cuda").manual_seed(7 * lvl + ci)
            w = torch.randn(27, ci, co, device="cuda", generator=gen) / np.sqrt(ci * 9)
            def feats(c):
                co_ = c.level_coords(lvl).float()[:, 1:]
                return torch.sin(co_ @ torch.tensor([[0.013], [0.007], [0.019]], device="cuda") + torch.arange(ci, device="cuda") * 0.37).contiguous()
            ya = alone.sparse_conv(kind, lvl, feats(alone), w)
            yb = batch.sparse_conv(kind, lvl, feats(batch), w)
            cb = batch.level_coords(lvl)
            rows = (cb[:, 0] == which).nonzero().squeeze(1)
            d = (yb[rows] - ya).abs()
            bad = (d.max(1).values > 0).nonzero().squeeze(1)
            print(f"scan {which} L{lvl} {ci}->{co} kw={kw}: rows {len(rows)} differing {len(bad)} max {float(d.max()):.3e}", bad[:12].tolist(), flush=True)
