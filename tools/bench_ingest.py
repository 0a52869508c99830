#!/usr/bin/env python
"""PCIe-inclusive rate: raw host scans (n,4) float32 -> pinned staging -> device -> filter -> voxelise -> forward -> top-128,
next to the device-resident rate bench.py reports (same workload: 16 scans x 50k returns, Cartesian 0.1 m)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g; g.build()
import egonn_amd
from egonn_amd.ingest import ScanIngest
from egonn_amd.synth import lidar_scan, seeded_state_dict

dev = torch.device("cuda", 0)
mp = egonn_amd.ModelParams(model="egonn", coordinates="cartesian", quantization_step=0.1)
model = egonn_amd.model_factory(mp)
sd = seeded_state_dict(1, {k: tuple(v.shape) for k, v in model.state_dict().items()})
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
model = model.to(dev).eval(); model.coord_bits = 12
ex = egonn_amd.DescriptorExtractor(model, n_k=128)
raws = []
for i in range(16):
    pc = lidar_scan(1000 + i, n_points=50000)
    raws.append(np.ascontiguousarray(np.concatenate([pc, np.ones((len(pc), 1), np.float32)], 1)))
ing = ScanIngest("mulran", dev, remove_ground_plane=False)
def step():
    pts, off = ing(raws)
    return ex.extract_packed(pts, off)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): ing(raws)
torch.cuda.synchronize(); t_ing = (time.perf_counter() - t0) / 20
pts, off = ing(raws)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): ex.extract_packed(pts, off)
torch.cuda.synchronize(); t_ext = (time.perf_counter() - t0) / 20
print(json.dumps({"ingest_ms": round(t_ing * 1e3, 3), "extract_ms": round(t_ext * 1e3, 3)}))
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 20
for _ in range(K): out = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print(json.dumps({"pcie_inclusive_scans_per_s": round(16 / dt, 1), "ms_per_step": round(dt * 1e3, 3),
                  "raw_MB_per_step": round(sum(r.nbytes for r in raws) / 1e6, 2), "note": "single batch in flight, host staging copy included"}))
