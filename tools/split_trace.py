#!/usr/bin/env python
"""Where does a wave of the lock-step split-bf16 kernel (sconv_split_kernel) spend its time?  s_memtime stamps per wave task:
   t0 start | t1 tables in LDS | t2 first requests landed | t3 first barrier passed | sums over the step loop of
   issue / compute / drain (s_waitcnt vmcnt(0)) / barrier ticks | steps with arithmetic | t9 loop end | t10 stores drained | steps
    B=16 LAYER=1 NW=4 python tools/split_trace.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g; g.build()
from egonn_amd import _lib
from egonn_amd.synth import lidar_scan
B = int(os.environ.get("B", 16)); NW = int(os.environ.get("NW", 4))
scans = [lidar_scan(1000 + i, 50000) for i in range(B)]
off = [0]
for s in scans: off.append(off[-1] + len(s))
pts = torch.from_numpy(np.concatenate(scans)).cuda()
ctx = _lib.Context(coord_bits=12)
ctx.voxelize(pts, off, 0, [0.1])
cfgs = [(1, 1, 32, 32), (0, 1, 32, 32), (1, 2, 32, 32), (0, 2, 32, 64), (0, 2, 64, 64), (1, 3, 64, 64), (0, 3, 64, 64),
        (0, 4, 128, 128), (0, 5, 128, 128), (0, 6, 128, 128), (0, 7, 128, 128)]
kind, lvl, ci, co = cfgs[int(os.environ.get("LAYER", 1))]
lin = lvl if kind == 0 else (lvl - 1 if kind == 1 else lvl + 1)
K = 27 if kind == 0 else 8
x = torch.randn(ctx.level_count(lin), ci, device="cuda")
w = torch.randn(K, ci, co, device="cuda") * 0.05
ng = ctx.map_groups(kind, lvl)[0]
buf = torch.zeros((ng + 64, 12), dtype=torch.int64, device="cuda")
var = 1000 + 9000 + 100 + NW * 10 + 2
ctx.lib.egonn_debug_set_naive_conv(ctx.h, var)
for _ in range(3): ctx.sparse_conv(kind, lvl, x, w)
buf.zero_()
ctx.lib.egonn_debug_set_trace(buf.data_ptr())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ctx.sparse_conv(kind, lvl, x, w); e1.record(); torch.cuda.synchronize()
ctx.lib.egonn_debug_set_trace(None)
ctx.lib.egonn_debug_set_naive_conv(ctx.h, 0)
t = buf.cpu().numpy()[:ng].astype(np.float64)
t = t[t[:, 10] > 0]
t0 = t[:, 0].min(); span = t[:, 10].max() - t0
print(f"kind {kind} L{lvl} {ci}->{co} NW={NW}: {len(t)} wave tasks, event {e0.elapsed_time(e1) * 1e3:.1f} us (incl. pack), span {span:.0f} ticks "
      f"({e0.elapsed_time(e1) * 1e6 / span:.2f} ns/tick if the span were the event)")
def row(name, v):
    print(f"  {name:34s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p50 {np.percentile(v, 50):9.0f}  p90 {np.percentile(v, 90):9.0f}")
row("tables (t0->t1)", t[:, 1] - t[:, 0])
row("first requests (t1->t2)", t[:, 2] - t[:, 1])
row("first barrier (t2->t3)", t[:, 3] - t[:, 2])
row("step loop (t3->t9)", t[:, 9] - t[:, 3])
row("store drain (t9->t10)", t[:, 10] - t[:, 9])
row("whole task", t[:, 10] - t[:, 0])
st = t[:, 11]
row("steps", st); row("steps with arithmetic", t[:, 8])
for name, c in [("issue", 4), ("compute", 5), ("drain vmcnt(0)", 6), ("barrier", 7)]:
    print(f"  per step: {name:16s} {t[:, c].sum() / st.sum():8.0f} ticks   (per step with arithmetic: {t[:, c].sum() / max(t[:, 8].sum(), 1):8.0f})")
# the XCDs' counters are not synchronised: cluster the tasks by clock domain (start stamps far apart) and look inside each
order = np.argsort(t[:, 0]); ts = t[order]
cuts = np.nonzero(np.diff(ts[:, 0]) > 1e7)[0] + 1
for ci_, cl in enumerate(np.split(ts, cuts)):
    c0 = cl[:, 0].min(); sp = cl[:, 10].max() - c0
    bins = np.linspace(0, sp, 21)
    alive = [int((((cl[:, 0] - c0) < hi) & ((cl[:, 10] - c0) > lo)).sum()) for lo, hi in zip(bins[:-1], bins[1:])]
    started = [int(((cl[:, 0] - c0) < hi).sum()) for hi in bins[1:]]
    print(f"  clock domain {ci_}: {len(cl)} wave tasks, span {sp:.0f} ticks = {sp / 2.4e3:.1f} us at 2.4 GHz; mean concurrency {((cl[:, 10] - cl[:, 0]).sum() / sp):.0f} waves")
    print("     alive per 1/20 span:", alive)
    print("     started by        :", started)
