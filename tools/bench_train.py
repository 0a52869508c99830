#!/usr/bin/env python
"""Training-step timing on one MI355X (per-GPU share of BASELINE configs[3]: 32 scans per rank).

    python tools/bench_train.py [--batch 32] [--points 50000] [--steps 10] [--polar] [--table out.json]

One step = forward (batch-statistics BN) + batch-hard triplet loss + backward + Adam (reference training/trainer.py:
157-175,193 with config/config_egonn.txt: batch 32, lr 1e-3, weight decay 1e-4, margin 0.2).  Prints ms/step and, with
--table, the per-kernel breakdown from torch.profiler."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

p = argparse.ArgumentParser()
p.add_argument("--batch", type=int, default=32)
p.add_argument("--points", type=int, default=50000)
p.add_argument("--steps", type=int, default=10)
p.add_argument("--polar", action="store_true")
p.add_argument("--table", default="")
args = p.parse_args()

import __graft_entry__ as ge
ge.build()
import egonn_amd
from egonn_amd.synth import lidar_scan, seeded_state_dict
from egonn_amd.train import TrainStep

dev = torch.device("cuda", 0)
if args.polar:
    mp = egonn_amd.ModelParams(model="egonn", coordinates="polar", quantization_step=[1.0, 0.3, 0.2])
else:
    mp = egonn_amd.ModelParams(model="egonn", coordinates="cartesian", quantization_step=0.1)
model = egonn_amd.model_factory(mp)
sd = seeded_state_dict(1, {k: tuple(v.shape) for k, v in model.state_dict().items()})
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
model = model.to(dev)
model.coord_bits = 12 if not args.polar else 16
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
step = TrainStep(model, opt, margin=0.2)

B = args.batch
coords = []
for b in range(B):
    c, _ = mp.quantizer(torch.from_numpy(lidar_scan(b, n_points=args.points)).to(dev))
    coords.append(torch.cat([torch.full((len(c), 1), b, dtype=torch.int32, device=dev), c.to(torch.int32)], 1))
coords = torch.cat(coords)
batch = {"coords": coords, "features": torch.ones((len(coords), 1), device=dev), "batch_size": B}
# consecutive scans are positives of each other in pairs, everything else negative
idx = torch.arange(B)
pos = (idx[:, None] // 2 == idx[None, :] // 2) & (idx[:, None] != idx[None, :])
neg = idx[:, None] // 2 != idx[None, :] // 2

for _ in range(2):
    loss, stats = step(batch, pos, neg)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    loss, stats = step(batch, pos, neg)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
res = {"ms_per_step": round(dt * 1e3, 2), "scans_per_s": round(B / dt, 1), "batch": B, "voxels": int(len(coords)),
       "loss": float(loss), "num_triplets": stats["num_triplets"], "quantizer": "polar" if args.polar else "cartesian 0.1 m"}
print(json.dumps(res))
if args.table:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(3):
            step(batch, pos, neg)
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages():
        t = getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)
        if t > 0 and e.device_type.name != "CPU":
            rows.append({"kernel": e.key[:90], "calls_per_step": e.count / 3, "us_per_step": t / 3})
    rows.sort(key=lambda r: -r["us_per_step"])
    json.dump({"summary": res, "kernels": rows[:40]}, open(args.table, "w"), indent=1)
    for r in rows[:25]:
        print(f"{r['us_per_step']:10.1f} us  x{r['calls_per_step']:6.1f}  {r['kernel']}")
