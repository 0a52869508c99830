#!/usr/bin/env python
"""One table of the other BASELINE.json workloads on one MI355X (profiles/r01u_workloads.json).  Synthetic clouds,
seeded random weights; every row says what a "scan" and a "step" is.  bench.py stays the headline (configs[1])."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g; g.build()
import egonn_amd
from egonn_amd import retrieval
from egonn_amd.ingest import ScanIngest
from egonn_amd.distributed import DatabaseBuilder
from egonn_amd.synth import lidar_scan, seeded_state_dict

dev = torch.device("cuda", 0)
rows = []

def make(model="egonn", coordinates="cartesian", step=0.1, **kw):
    mp = egonn_amd.ModelParams(model=model, coordinates=coordinates, quantization_step=step, **kw)
    m = egonn_amd.model_factory(mp)
    sd = seeded_state_dict(1, {k: tuple(v.shape) for k, v in m.state_dict().items()})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return mp, m.to(dev).eval()

def timeit(fn, n, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

# configs[0]-shaped: ONE 120k-point scan, KITTI-style filtered, polar 1 deg / 0.3 m / 0.2 m and Cartesian 0.3 m (latency)
pc = lidar_scan(1, n_points=120000); pc = pc[pc[:, 2] > -1.5]
for coords, step in (("polar", [1.0, 0.3, 0.2]), ("cartesian", 0.3)):
    mp, m = make(coordinates=coords, step=step)
    ex = egonn_amd.DescriptorExtractor(m, n_k=128)
    p = torch.from_numpy(pc).to(dev)
    dt = timeit(lambda: ex.extract_packed(p, [0, len(p)]), 50)
    rows.append({"workload": f"configs[0] shape: one {len(pc)}-pt scan, {coords} {step}", "ms_per_scan": round(dt * 1e3, 3),
                 "voxels": m.context().level_count(0)})

# configs[2]: bf16 operands, batch 64, 3 batches in flight
mp, m = make(); m.coord_bits = 12; m.precision = "bf16"
ex = egonn_amd.DescriptorExtractor(m, n_k=128)
scans = [lidar_scan(1000 + i, 50000) for i in range(64)]
off = [0]
for s in scans: off.append(off[-1] + len(s))
pts = torch.from_numpy(np.concatenate(scans)).to(dev)
def run_stream(k, n_streams=3):
    for o in ex.extract_stream(((pts, off) for _ in range(k)), n_streams=n_streams): pass
run_stream(6); torch.cuda.synchronize(); t0 = time.perf_counter(); run_stream(12); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 12
rows.append({"workload": "configs[2]: bf16 MFMA operands, batch 64 x 50k pts, Cartesian 0.1 m, 3 batches in flight",
             "scans_per_s": round(64 / dt, 1), "ms_per_step": round(dt * 1e3, 3)})
m.precision = "fp32"
run_stream(6); torch.cuda.synchronize(); t0 = time.perf_counter(); run_stream(12); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 12
rows.append({"workload": "same, fp32", "scans_per_s": round(64 / dt, 1), "ms_per_step": round(dt * 1e3, 3)})

# MinkLoc3D (second model family), batch 16 x 50k pts, Cartesian 0.1 m... the reference config uses 0.01 normalised clouds; here metres
mp, m = make(model="MinkLoc3D", step=0.1); m.coord_bits = 12
c16 = []
for b in range(16):
    c, _ = mp.quantizer(torch.from_numpy(scans[b]).to(dev))
    c16.append(torch.cat([torch.full((len(c), 1), b, dtype=torch.int32, device=dev), c.to(torch.int32)], 1))
c16 = torch.cat(c16); f16 = torch.ones((len(c16), 1), device=dev)
dt = timeit(lambda: m({"coords": c16, "features": f16, "batch_size": 16}), 20)
rows.append({"workload": "MinkLoc3D forward, batch 16 (pre-quantised coordinates)", "scans_per_s": round(16 / dt, 1), "ms_per_step": round(dt * 1e3, 3)})

# configs[4]-shaped database build on one GPU: 2000 MulRan-shaped raw scans (65 536 returns, x y z reflectance) ->
# device ingest (ground cut z > -0.9) -> descriptors -> kNN(20) + recall over 500 queries
mp, m = make(); m.coord_bits = 12
ex = egonn_amd.DescriptorExtractor(m, n_k=128)
ing = ScanIngest("mulran", dev)
raw = []
for i in range(16):
    s = lidar_scan(3000 + i, 65536)
    raw.append(np.ascontiguousarray(np.concatenate([s, np.ones((len(s), 1), np.float32)], 1)))
def db_batch():
    p, o = ing(raw)
    return ex.extract_packed(p, o)["global"]
n_batches = 125                                    # 2000 scans
for _ in range(3): db_batch()
torch.cuda.synchronize(); t0 = time.perf_counter()
G = torch.cat([db_batch() for _ in range(n_batches)])
torch.cuda.synchronize(); t_build = time.perf_counter() - t0
G = G + 1e-3 * torch.randn_like(G)                 # distinct rows (the 16 synthetic scans repeat)
pos = torch.rand((len(G), 2), device=dev) * 1000
t0 = time.perf_counter()
out = retrieval.recall_at_k(G, G[:500] + 1e-4, pos, pos[:500], radius=[5, 20], k=20)
torch.cuda.synchronize(); t_knn = time.perf_counter() - t0
rows.append({"workload": "configs[4] shape on ONE GPU: 2000 raw MulRan-shaped scans -> ingest (host staging + H2D + filter) -> "
                         "descriptors (batch 16, one batch in flight) ; then kNN(20)+recall of 500 queries over the 2000 descriptors",
             "build_scans_per_s": round(len(G) / t_build, 1), "knn_recall_ms": round(t_knn * 1e3, 2),
             "recall@1_r5": out["recall"][5][0]})
json.dump(rows, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "workloads.json"), "w"), indent=1)
for r in rows: print(r)
