#!/usr/bin/env python
"""Where does a wave of the sparse-conv kernel spend its time?  Runs the traced build of the LDS-DMA kernel on one layer of
the benchmark batch and summarises the per-task s_memtime stamps:
   t0 task start | t1 tables in LDS | t2 first D-1 items issued | t3 last MFMA issued | t4 epilogue done
    B=16 LAYER=1 python tools/sconv_trace.py        (LAYER indexes the cfgs list of tools/bench_sconv.py)"""
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
cfgs = [(1, 1, 32, 32), (0, 1, 32, 32), (1, 2, 32, 32), (0, 2, 32, 64), (0, 2, 64, 64), (1, 3, 64, 64), (0, 3, 64, 64),
        (0, 4, 128, 128), (0, 5, 128, 128), (0, 6, 128, 128), (0, 7, 128, 128)]      # 128->128: the traced split-bf16 per-tile kernel
kind, lvl, ci, co = cfgs[int(os.environ.get("LAYER", 1))]
lin = lvl if kind == 0 else (lvl - 1 if kind == 1 else lvl + 1)
K = 27 if kind == 0 else 8
n_in = ctx.level_count(lin)
x = torch.randn(n_in, ci, device="cuda")
w = torch.randn(K, ci, co, device="cuda") * 0.05
ng = ctx.map_groups(kind, lvl)[0]
ksp = min(ci // 32, 4)
ntr = ng * (co // 32) * ksp
buf = torch.zeros((ntr + 64, 8), dtype=torch.int64, device="cuda")
ctx.lib.egonn_debug_set_naive_conv(ctx.h, 2009 if ci == 128 else 128)
for _ in range(3): ctx.sparse_conv(kind, lvl, x, w)
buf.zero_()
ctx.lib.egonn_debug_set_trace(buf.data_ptr())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ctx.sparse_conv(kind, lvl, x, w); e1.record(); torch.cuda.synchronize()
ctx.lib.egonn_debug_set_trace(None)
ctx.lib.egonn_debug_set_naive_conv(ctx.h, 0)
t = buf.cpu().numpy()[:ntr]
t = t[t[:, 4] > 0]
t0 = t[:, 0].min()
span = t[:, 4].max() - t0
d = lambda a, b: (t[:, b] - t[:, a]).astype(np.float64)
print(f"layer kind {kind} L{lvl} {ci}->{co}: {len(t)} wave tasks, event time {e0.elapsed_time(e1) * 1e3:.1f} us (incl. pack), span {span} ticks")
tick_ns = e0.elapsed_time(e1) * 1e6 / span
print(f"  (if the span were the whole event: {tick_ns:.2f} ns per tick)")
items = t[:, 5].astype(np.float64)
for name, a, b in [("tables (t0->t1)", 0, 1), ("prologue issue (t1->t2)", 1, 2), ("item loop (t2->t3)", 2, 3), ("epilogue (t3->t4)", 3, 4), ("whole task", 0, 4)]:
    x_ = d(a, b)
    print(f"  {name:26s} mean {x_.mean():9.0f}  p10 {np.percentile(x_, 10):9.0f}  p50 {np.percentile(x_, 50):9.0f}  p90 {np.percentile(x_, 90):9.0f} ticks")
loop = d(2, 3)
print(f"  items per task mean {items.mean():.1f}; item loop ticks per item {loop.sum() / items.sum():.0f}")
if ci == 128:      # traced split build: the last two words carry the phase sums of the pipelined item loop
    ph = [(t[:, 6] >> 32) & 0xFFFFFFFF, t[:, 6] & 0xFFFFFFFF, (t[:, 7] >> 32) & 0xFFFFFFFF, t[:, 7] & 0xFFFFFFFF]
    nm = (t[:, 5] // 4) * 4
    for name, v in zip(("issue (8 loads) + generate", "vmcnt wait + LDS read issue", "split + 12 MFMA issue", "LDS wait"), ph):
        print(f"  per pipelined item: {name:30s} {v.sum() / max(nm.sum(), 1):7.0f} ticks")
    sys.exit(0)
# occupancy over time: wave tasks alive per tick-bin, and per SIMD
hw = t[:, 6].astype(np.int64)
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7   # HW_ID fields (gfx9 layout)
bins = np.linspace(0, span, 41)
alive = [(((t[:, 0] - t0) < hi) & ((t[:, 4] - t0) > lo)).sum() for lo, hi in zip(bins[:-1], bins[1:])]
print("  wave tasks alive per 1/40 of the span:", alive)
starts = np.sort(t[:, 0] - t0)
print("  task start times p10/p50/p90/max of span: %.2f %.2f %.2f %.2f" % tuple(np.percentile(starts, [10, 50, 90, 100]) / span))
if len(sys.argv) > 1:
    np.save(sys.argv[1], t)

# per-CU view (s_memtime is only comparable within a CU)
xcd = t[:, 7] & 7
key = (xcd << 20) | (((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5))
spans, occ, mf = [], [], []
for k_ in np.unique(key):
    tt = t[key == k_]
    sp = tt[:, 4].max() - tt[:, 0].min()
    spans.append(sp); occ.append((tt[:, 4] - tt[:, 0]).sum() / sp); mf.append((tt[:, 5] * 16 * 32).sum() / (4 * sp))
print("  per CU: span cycles p10/50/90/max", np.percentile(spans, [10, 50, 90, 100]).astype(int).tolist(),
      " live waves (avg over the span) p10/50/90 %.1f %.1f %.1f" % tuple(np.percentile(occ, [10, 50, 90])),
      " MFMA issue share of the span p10/50/90 %.2f %.2f %.2f" % tuple(np.percentile(mf, [10, 50, 90])))
