#!/usr/bin/env python
"""Is the single host thread the limiter of the streamed throughput mode?  One Python thread per stream
(ctypes releases the GIL inside libegonn_hip calls)."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g; g.build()
from egonn_amd import ModelParams, model_factory, DescriptorExtractor
from egonn_amd.synth import lidar_scan, seeded_state_dict
S = int(os.environ.get("S", 3)); STEPS = int(os.environ.get("STEPS", 60))
mp = ModelParams(model="egonn", coordinates="cartesian", quantization_step=0.1)
model = model_factory(mp)
sd = seeded_state_dict(1, {k: tuple(v.shape) for k, v in model.state_dict().items()})
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
model = model.to("cuda").eval(); model.coord_bits = 12
scans = [lidar_scan(1000 + i, 50000) for i in range(16)]
off = [0]
for s in scans: off.append(off[-1] + len(s))
pts = torch.from_numpy(np.concatenate(scans)).cuda()
exs = [DescriptorExtractor(model, 128) for _ in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]
model._sync_weights()
for i in range(S):
    with torch.cuda.stream(streams[i]):
        for _ in range(3): exs[i].extract_packed(pts, off, slot=i)
torch.cuda.synchronize()
def work(i, n):
    with torch.cuda.stream(streams[i]):
        for _ in range(n): exs[i].extract_packed(pts, off, slot=i)
    streams[i].synchronize()
t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(i, STEPS // S)) for i in range(S)]
for t in th: t.start()
for t in th: t.join()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"threads={S} steps={STEPS // S * S} scans/s={16 * (STEPS // S * S) / dt:.0f} ms/step={dt / (STEPS // S * S) * 1e3:.3f}")
