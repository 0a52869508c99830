"""A/B of the resident tail kernel (csrc/tail.hip) against the per-layer launches: deviation of the level-5..7 maps and of the
global descriptor, and the time of one eager step (one batch in flight) with either path."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
import egonn_amd as E
from egonn_amd.synth import lidar_scan, seeded_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mp = E.ModelParams(model="egonn", coordinates="cartesian", quantization_step=0.1)
m = E.model_factory(mp)
shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
m.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_state_dict(7, shapes).items()})
m = m.to("cuda").eval()
m.coord_bits = 12
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 1000      # 1000: the first batch of bench.py
scans = [lidar_scan(SEED + i, 50000) for i in range(B)]
off = [0]
for s in scans:
    off.append(off[-1] + len(s))
pts = torch.from_numpy(np.concatenate(scans)).cuda()
ex = E.DescriptorExtractor(m, n_k=128)
ctx = m.context(0)
res = {}
for mode in (1, 0):
    ctx.set_tail(mode)
    out = ex.extract_packed(pts, off)
    res[mode] = (out["global"].clone(), [ctx.forward_level_features(l, 128).clone() for l in (5, 6, 7)])
    ctx.plan_status()
    torch.cuda.synchronize()
    for _ in range(5):
        ex.extract_packed(pts, off)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        ex.extract_packed(pts, off)
    torch.cuda.synchronize()
    print(f"tail_mode {mode}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per eager step (batch {B})")
    # exclusive durations (HIP events around every tagged launch) of what the tail replaces / of the tail kernel
    ctx.profile_enable(1)
    ctx.profile_fetch()
    for _ in range(5):
        ex.extract_packed(pts, off)
    recs = ctx.profile_fetch()
    ctx.profile_enable(0)
    agg = {}
    for nm, ms, by, fl in recs:
        agg.setdefault(nm, []).append(ms * 1e3)
    tot_all = sum(np.mean(v) for v in agg.values())
    sel = {k: np.mean(v) for k, v in agg.items() if any(x in k for x in ("/L5", "/L6", "/L7", "tail_", "gl_", "global", "gem", "pool", "g1x1", "gt", "gdec"))}
    for k, v in sorted(sel.items(), key=lambda kv: -kv[1]):
        print(f"   {k:56s} {v:8.1f} us")
    print(f"   tagged launches: all {tot_all:.1f} us, listed {sum(sel.values()):.1f} us")
g1, f1 = res[1]
g0, f0 = res[0]
for l, a, b in zip((5, 6, 7), f0, f1):
    print(f"level {l}: rows {a.shape[0]}, max|diff|/max|ref| = {float((a - b).abs().max() / b.abs().max()):.3e}, equal = {torch.equal(a, b)}")
print(f"global: max|diff|/max|ref| = {float((g0 - g1).abs().max() / g1.abs().max()):.3e}, equal = {torch.equal(g0, g1)}")
