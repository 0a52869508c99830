"""probe: every map kind / level of the polar training fixture's plan, product rule vs plain kernel"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import __graft_entry__ as g; g.build()
from egonn_amd import _lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import helpers as H
case = H.load_case(sys.argv[1] if len(sys.argv) > 1 else "egonn_train_polar")
coords = torch.from_numpy(case["coords"]).cuda()
B = int(coords[:, 0].max().item()) + 1
def plan():
    c = _lib.Context(coord_bits=12); c.set_coords(coords, B) if hasattr(c, "set_coords") else c.coords_set(coords, B); return c
ctx, ref = plan(), plan()
ref.set_naive_conv(True)
ctx.prepare_maps(True); ref.prepare_maps(True)
print("levels", [ctx.level_count(l) for l in range(8)])
chan = [32, 32, 64, 64, 128, 128, 128, 128]
for kind in (0, 1, 2):
    for lvl in range(0 if kind == 2 else 1, 7 if kind == 2 else 8):
        lin = lvl if kind == 0 else (lvl - 1 if kind == 1 else lvl + 1)
        K = 27 if kind == 0 else 8
        for (ci, co) in {(chan[lin], chan[lvl]), (64, 64), (128, 128), (64, 128), (128, 64), (64, 32)}:
            if ctx.level_count(lin) == 0 or ctx.level_count(lvl) == 0: continue
            x = torch.randn(ctx.level_count(lin), ci, device="cuda")
            w = torch.randn(K, ci, co, device="cuda") * 0.05
            try:
                want = ref.sparse_conv(kind, lvl, x, w); got = ctx.sparse_conv(kind, lvl, x, w)
            except Exception as e:
                print("kind", kind, "L", lvl, ci, co, "ERR", str(e)[:100]); continue
            err = float((got - want).abs().max() / (want.abs().max() + 1e-9))
            if err > 3e-6: print(f"kind {kind} L{lvl} {ci}->{co}: rel err {err:.3e} rows {ctx.level_count(lvl)} BAD", flush=True)
print("done")
