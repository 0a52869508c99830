"""Per-stage s_memtime stamps of the resident tail kernel (csrc/tail.hip; egonn_debug_set_trace): where a workgroup's time
goes — waiting for its cluster, staging rounds, items (MFMA loops), reduction + epilogue.  usage: tail_trace.py [batch]"""
import os, sys
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
ctx.set_tail(0)
for _ in range(3):
    ex.extract_packed(pts, off)
torch.cuda.synchronize()
buf = torch.zeros((1 << 20,), dtype=torch.int64, device="cuda")
ctx.lib.egonn_debug_set_trace(buf.data_ptr())
ex.extract_packed(pts, off)
torch.cuda.synchronize()
ctx.lib.egonn_debug_set_trace(None)
t = buf[(1 << 19):(1 << 19) + B * 8 * 17 * 8].cpu().numpy().reshape(B * 8, 17, 8).astype(np.float64)
names = ["L5 k2s2", "L5 conv1", "L5 conv2", "L5 gate", "L6 k2s2", "L6 conv1", "L6 conv2", "L6 gate", "L7 k2s2", "L7 conv1",
         "L7 conv2", "L7 gate", "H7 1x1", "H6", "H5", "MLP1", "MLP2+pool"]
t0 = t[:, 0, 0].min()
print(f"rows per scan at level 5/6/7: {[ctx.level_count(l) / B for l in (5, 6, 7)]}; ticks are s_memtime units")
print(f"kernel span (first start .. last end): {(t[:, :, 5].max() - t0):.0f} ticks")
tot = [0.0] * 6
print(f"{'stage':12s} {'span':>8s} {'wait':>8s} {'stage':>8s} {'items':>8s} {'red+epi':>8s} {'rounds':>6s} {'W wait':>8s}   (mean over workgroups; span = start..published)")
for s, nm in enumerate(names):
    span = (t[:, s, 5] - t[:, s, 0]).mean()
    wait = np.where(t[:, s, 1] > 0, t[:, s, 1] - t[:, s, 0], 0).mean()
    print(f"{nm:12s} {span:8.0f} {wait:8.0f} {t[:, s, 2].mean():8.0f} {t[:, s, 3].mean():8.0f} {t[:, s, 4].mean():8.0f} {t[:, s, 6].mean():6.1f} {t[:, s, 7].mean():8.0f}")
    for i, v in enumerate((span, wait, t[:, s, 2].mean(), t[:, s, 3].mean(), t[:, s, 4].mean(), t[:, s, 7].mean())):
        tot[i] += v
print(f"{'sum':12s} {tot[0]:8.0f} {tot[1]:8.0f} {tot[2]:8.0f} {tot[3]:8.0f} {tot[4]:8.0f} {'':6s} {tot[5]:8.0f}   other {tot[0] - sum(tot[1:5]):.0f}")
