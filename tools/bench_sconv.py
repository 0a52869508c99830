#!/usr/bin/env python
"""Micro-benchmark + correctness check of the sparse-conv kernel (sconv.hip) on the benchmark workload.

For every layer shape of the EgoNN trunk at batch B: time of the launch (HIP events over 20 launches), algorithmic
bytes, HBM-roofline fraction, MFMA-roofline fraction, padding factor of the row groups, and the max
error against the plain (one thread per output) kernel.  fp32 and bf16 feature maps.
    B=16 python tools/bench_sconv.py [out.json]
"""
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
ref = _lib.Context(coord_bits=12)
ref.voxelize(pts, off, 0, [0.1])
ref.set_naive_conv(True)
print("levels", [ctx.level_count(l) for l in range(8)], flush=True)

# (map kind, out level, cin, cout)
cfgs = [(1, 1, 32, 32), (0, 1, 32, 32), (1, 2, 32, 32), (0, 2, 32, 64), (0, 2, 64, 64), (1, 3, 64, 64), (0, 3, 64, 64),
        (1, 4, 64, 64), (0, 4, 64, 128), (0, 4, 128, 128), (1, 5, 128, 128), (0, 5, 128, 128), (0, 6, 128, 128),
        (0, 7, 128, 128), (2, 6, 128, 128), (2, 5, 128, 128), (2, 3, 64, 64)]
if os.environ.get("ONLY"):
    cfgs = [cfgs[int(i)] for i in os.environ["ONLY"].split(",")]
rows = []
for (kind, lvl, ci, co) in cfgs:
    lin = lvl if kind == 0 else (lvl - 1 if kind == 1 else lvl + 1)
    K = 27 if kind == 0 else 8
    n_in, n_out = ctx.level_count(lin), ctx.level_count(lvl)
    torch.manual_seed(lvl * 100 + ci)
    x = torch.randn(n_in, ci, device="cuda")
    w = torch.randn(K, ci, co, device="cuda") * (1.0 / np.sqrt(ci * (9 if K == 27 else 2)))
    # pairs of the map = non-zero rows gathered: count with an all-ones kernel on all-ones features
    quick = bool(os.environ.get("QUICK"))       # profiling runs: only the timed launches
    P = 1.0
    if not quick:
        ones = ctx.sparse_conv(kind, lvl, torch.ones(n_in, 32, device="cuda"), torch.ones(K, 32, 32, device="cuda") / 32)
        P = float(ones[:, 0].sum().item())
    want = ref.sparse_conv(kind, lvl, x, w) if (n_out * co * K < 6e8 and not quick) else None
    ng = ctx.map_groups(kind, lvl)[0]
    variants = [int(v) for v in os.environ["AB"].split(",")] if os.environ.get("AB") else [0]
    dts = (torch.float32,) if os.environ.get("F32ONLY") else (torch.float32, torch.bfloat16)
    for dt, var in [(d, v) for d in dts for v in variants]:
        if var >= 1000 and dt != torch.float32: continue
        ctx.lib.egonn_debug_set_naive_conv(ctx.h, var)
        xx = x.to(dt).contiguous()
        got = ctx.sparse_conv(kind, lvl, xx, w)
        err = float("nan")
        if want is not None:
            err = float(((got.float() - want).abs().max() / (want.abs().max() + 1e-9)).item())
        for _ in range(3): ctx.sparse_conv(kind, lvl, xx, w)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ctx.sparse_conv(kind, lvl, xx, w)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        # the per-op entry repacks the kernel on every call: time that alone and subtract
        es = 4 if dt == torch.float32 else 2
        alg = P * ci * es + n_out * co * es + K * ci * co * es + 8 * P
        fl = 2 * P * ci * co
        ctx.lib.egonn_debug_set_naive_conv(ctx.h, 0)
        rows.append(dict(kind=kind, level=lvl, cin=ci, cout=co, dtype=str(dt).split(".")[-1] + ("/v%d" % var if var else ""), us=us, pairs=P, n_out=n_out,
                         groups=ng, alg_MB=alg / 1e6, hbm_frac=alg / (us * 1e-6) / 8e12, tflops=fl / (us * 1e-6) / 1e12, rel_err=err))
        r = rows[-1]
        print(f"kind {kind} L{lvl} {ci:3d}->{co:3d} {r['dtype']:16s}: {us:8.1f} us  alg {r['alg_MB']:7.1f} MB  hbm_frac {r['hbm_frac']:.3f}  "
              f"{r['tflops']:6.1f} TF  err {err:.2e}  pairs/row {P / max(n_out, 1):.2f} groups {ng}", flush=True)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
