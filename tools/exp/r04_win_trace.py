#!/usr/bin/env python
"""Where does a wave of the window-resident kernel (sconv_win_kernel) spend its time?  s_memtime stamps per wave:
   t0 start | t1 round trip 1 issued + staging issued | t2 staged rows landed | t3 barrier passed | t4 step loop end | t5 stores drained
   | offsets in the wave's union, overflow flag, halo rows | present (group, offset) pairs
    B=16 LAYER=1 G=4 python tools/win_trace.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g; g.build()
from egonn_amd import _lib
from egonn_amd.synth import lidar_scan
B = int(os.environ.get("B", 16)); G = int(os.environ.get("G", 4))
scans = [lidar_scan(1000 + i, 50000) for i in range(B)]
off = [0]
for s in scans: off.append(off[-1] + len(s))
pts = torch.from_numpy(np.concatenate(scans)).cuda()
ctx = _lib.Context(coord_bits=12)
ctx.lib.egonn_debug_set_naive_conv(ctx.h, 7003)      # the window-resident kernel is off in the product
ctx.voxelize(pts, off, 0, [0.1])
cfgs = [None, (0, 1, 32, 32), None, None, (0, 2, 64, 64)]
kind, lvl, ci, co = cfgs[int(os.environ.get("LAYER", 1))]
x = torch.randn(ctx.level_count(lvl), ci, device="cuda")
w = torch.randn(27, ci, co, device="cuda") * 0.05
ng = ctx.map_groups(kind, lvl)[0]
NW = 16 // G
nwin = (ng + 15) // 16
buf = torch.zeros((nwin * NW + 64, 8), dtype=torch.int64, device="cuda")
ctx.lib.egonn_debug_set_naive_conv(ctx.h, 5900 + G)
for _ in range(3): ctx.sparse_conv(kind, lvl, x, w)
buf.zero_()
ctx.lib.egonn_debug_set_trace(buf.data_ptr())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ctx.sparse_conv(kind, lvl, x, w); e1.record(); torch.cuda.synchronize()
ctx.lib.egonn_debug_set_trace(None)
ctx.lib.egonn_debug_set_naive_conv(ctx.h, 0)
t = buf.cpu().numpy()[: nwin * NW]
t = t[t[:, 5] > 0]
steps = (t[:, 6] & 0xFFFFFFFF).astype(np.float64); direct = ((t[:, 6] >> 32) & 0xFFFF) > 0; halo = (t[:, 6] >> 48).astype(np.float64)
present = t[:, 7].astype(np.float64)
t = t.astype(np.float64)
print(f"L{lvl} {ci}->{co} G={G}: {len(t)} waves of {nwin} windows, event {e0.elapsed_time(e1) * 1e3:.1f} us (incl. pack); windows with halo rows beyond the slots {int(direct.sum()) // NW}")
def row(name, v):
    print(f"  {name:34s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p50 {np.percentile(v, 50):9.0f}  p90 {np.percentile(v, 90):9.0f}  max {v.max():9.0f}")
row("issue trip 1 + staging (t0->t1)", t[:, 1] - t[:, 0])
row("staged rows landed (t1->t2)", t[:, 2] - t[:, 1])
row("barrier (t2->t3)", t[:, 3] - t[:, 2])
row("step loop (t3->t4)", t[:, 4] - t[:, 3])
row("epilogue + drain (t4->t5)", t[:, 5] - t[:, 4])
row("whole wave", t[:, 5] - t[:, 0])
row("steps (offsets in the union)", steps); row("present (group, offset) pairs", present); row("halo rows", halo)
print(f"  loop ticks per step {((t[:, 4] - t[:, 3]).sum() / steps.sum()):.0f}, per present pair {((t[:, 4] - t[:, 3]).sum() / present.sum()):.0f}")
order = np.argsort(t[:, 0]); ts = t[order]
cuts = np.nonzero(np.diff(ts[:, 0]) > 1e7)[0] + 1
for ci_, cl in enumerate(np.split(ts, cuts)):
    c0 = cl[:, 0].min(); sp = cl[:, 5].max() - c0
    bins = np.linspace(0, sp, 21)
    alive = [int((((cl[:, 0] - c0) < hi) & ((cl[:, 5] - c0) > lo)).sum()) for lo, hi in zip(bins[:-1], bins[1:])]
    print(f"  clock domain {ci_}: {len(cl)} waves, span {sp:.0f} ticks; mean concurrency {((cl[:, 5] - cl[:, 0]).sum() / sp):.0f} waves; alive per 1/20 span: {alive}")
