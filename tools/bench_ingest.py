#!/usr/bin/env python
"""Host buffers -> descriptors (BASELINE configs[4] on ONE GPU): MulRan-shaped raw scans (65 536 returns, x y z reflectance
float32 = 1 MB per scan, datasets/mulran/mulran_raw.py:19-25) held in HOST memory -> pinned staging (reader threads) -> H2D ->
device filter (ground cut z > -0.9) -> voxelise -> forward -> top-128 -> global descriptors back on the host.
The PCIe-inclusive rate of the streaming pipeline (egonn_amd/stream.py) next to the eager per-batch path of round 2.

    python tools/bench_ingest.py [--scans 2000] [--slots 4] [--workers 8] [--json] [--files DIR]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
p = argparse.ArgumentParser()
p.add_argument("--scans", type=int, default=2000)
p.add_argument("--distinct", type=int, default=32, help="distinct synthetic scans (the stream cycles through them)")
p.add_argument("--slots", type=int, default=4)
p.add_argument("--workers", type=int, default=8)
p.add_argument("--batch", type=int, default=16)
p.add_argument("--keep-local", action="store_true")
p.add_argument("--files", default="", help="write the distinct scans as .bin files there and stream the FILES (readinto)")
p.add_argument("--json", action="store_true")
args = p.parse_args()
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
for i in range(args.distinct):
    pc = lidar_scan(3000 + i, n_points=65536, n_azimuth=1024)
    raws.append(np.ascontiguousarray(np.concatenate([pc, np.ones((len(pc), 1), np.float32)], 1)))
sources = raws
if args.files:
    os.makedirs(args.files, exist_ok=True)
    sources = []
    for i, r in enumerate(raws):
        fn = os.path.join(args.files, f"{i:06d}.bin"); r.tofile(fn); sources.append(fn)
B = args.batch
stream = [sources[i % len(sources)] for i in range(args.scans)]
batches = [stream[i:i + B] for i in range(0, len(stream), B)]

se = egonn_amd.StreamingExtractor(ex, batch_size=B, max_points_per_scan=65536, floats_per_point=4, dataset_type="mulran",
                                  slots=args.slots, workers=args.workers, keep_local=args.keep_local)
se.calibrate(raws[:B], margin=1.3)
for _ in se.run(batches[:2 * args.slots]): pass            # capture + warm-up
torch.cuda.synchronize(); t0 = time.perf_counter()
n_out = 0
for out in se.run(batches):
    n_out += out["global"].shape[0]
dt = time.perf_counter() - t0
assert n_out == args.scans
res = {"workload": f"{args.scans} MulRan-shaped raw scans (65 536 returns x 16 B) from host {'files' if args.files else 'buffers'} -> "
                   f"global descriptors on the host, batch {B}, {args.slots} batches in flight, {args.workers} reader threads"
                   + (", keypoints + local descriptors copied back too" if args.keep_local else ""),
       "scans_per_s": round(args.scans / dt, 1), "ms_per_batch": round(dt / len(batches) * 1e3, 3),
       "raw_GB_per_s": round(args.scans * 65536 * 16 / dt / 1e9, 2), "fallbacks": se.fallbacks}
# round-2 path for comparison: eager ingest (one pinned buffer, Python staging loop, host sync on the offsets) + eager extract
ing = ScanIngest("mulran", dev)
def eager(batch):
    arrs = [np.fromfile(s, dtype=np.float32).reshape(-1, 4) if isinstance(s, str) else s for s in batch]
    pts, off = ing(arrs)
    return ex.extract_packed(pts, off)["global"].cpu()
for b in batches[:3]: eager(b)
nb = min(len(batches), 20)
torch.cuda.synchronize(); t0 = time.perf_counter()
for b in batches[:nb]: eager(b)
dt2 = time.perf_counter() - t0
res["eager_scans_per_s"] = round(nb * B / dt2, 1)
print(json.dumps(res))
