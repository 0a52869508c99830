#!/usr/bin/env python
"""bench.py — descriptor-extraction throughput of the MI355X-native EgoNN path.

Metric (BASELINE.json): LiDAR scans/sec (descriptor extraction), 50k-pt clouds @ 0.1 m voxel.
Workload (BASELINE.json configs[1]): EgoNN inference, synthetic 50k-pt clouds, Cartesian 0.1 m voxels,
batch 16 per GPU, fp32 -> 256-d global descriptor + 128 keypoints + 128-d local descriptors.
A step = one pass of the whole hot path over one batch: voxelise -> forward -> top-128 selection, with the
points already resident in HBM when the timed region starts.  N>1: one process per GPU, each rank owns its
own batch (independent scans, no data-path collective) => weak scaling.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, HIP-event
timing inside the timed region) and `cpu_baseline` (the CPU oracle on a bounded sample, rank 0 at N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--batch", type=int, default=16, help="scans per GPU per step (BASELINE configs[1]: 16)")
    p.add_argument("--points", type=int, default=50_000)
    p.add_argument("--voxel", type=float, default=0.1)
    p.add_argument("--cpu-scans", type=int, default=6, help="scans timed on the numpy oracle (fallback only)")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU work given to the C/OpenMP oracle baseline")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--layer-table", type=str, default="", help="write a per-layer timing table (json) here")
    p.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                   help="f32 = BASELINE configs[1] (default, the headline metric); bf16 = configs[2] arithmetic "
                        "(sparse-conv MFMA operands rounded to bf16, fp32 accumulate, fp32 feature maps)")
    p.add_argument("--streams", type=int, default=3, help="batches in flight (HIP streams, one egonn_ctx each)")
    return p.parse_args()


def make_scans(rank: int, batch: int, n_points: int):
    from egonn_amd.synth import lidar_scan
    return [lidar_scan(1000 * rank + i, n_points=n_points) for i in range(batch)]


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)   # launched by torch.distributed.run
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    if distributed:
        dist.barrier()
    from egonn_amd import ModelParams, model_factory, DescriptorExtractor
    from egonn_amd.synth import seeded_state_dict

    dev = torch.device("cuda", torch.cuda.current_device())
    mp = ModelParams(model="egonn", coordinates="cartesian", quantization_step=args.voxel)
    model = model_factory(mp)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = seeded_state_dict(1, shapes)                      # random-init weights (no pretrained weights exist)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to(dev).eval()
    model.coord_bits = 12                                  # +-204.8 m at 0.1 m voxels: fewer radix passes
    model.precision = "bf16" if args.dtype == "bf16" else "fp32"
    ex = DescriptorExtractor(model, n_k=128)

    scans = make_scans(rank, args.batch, args.points)
    offsets = [0]
    for s in scans:
        offsets.append(offsets[-1] + len(s))
    points = torch.from_numpy(np.concatenate(scans, axis=0)).to(dev).contiguous()   # resident in HBM

    ctx = model.context()

    def step():
        return ex.extract_packed(points, offsets)

    # warm-up; the last warm-up step times every tagged launch to find the dominant kernel (largest total time)
    for i in range(max(args.warmup, 1)):
        if i == max(args.warmup, 1) - 1:
            ctx.profile_enable(1)
            ctx.profile_fetch()
        out = step()
    torch.cuda.synchronize()
    per_kernel = {}
    for name, t, b, f in ctx.profile_fetch():
        per_kernel[name.split("/")[0]] = per_kernel.get(name.split("/")[0], 0.0) + t
    dominant = max(per_kernel, key=per_kernel.get)
    ctx.profile_enable(2, dominant + "/")                  # HIP events around the dominant kernel's launches only

    # every in-flight slot needs its arenas grown before the timed region
    for o in ex.extract_stream(((points, offsets) for _ in range(2 * args.streams)), n_streams=args.streams):
        out = o
    torch.cuda.synchronize()
    ctx.profile_fetch()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = []
    for o in ex.extract_stream(((points, offsets) for _ in range(args.steps)), n_streams=args.streams):
        outs.append(o)
        if len(outs) > 2 * args.streams:
            outs.pop(0)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    recs = ctx.profile_fetch()
    # the same kernel with ONE batch in flight (after the timed region): rocprofv3 serialises dispatches of different
    # streams, so its per-kernel average is this "exclusive" duration, not the one measured while three batches
    # share the CUs
    excl = []
    if args.streams > 1:
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        excl = ctx.profile_fetch()
    ctx.profile_enable(0)
    n_levels = [ctx.level_count(l) for l in range(8)]

    # ---------------- roofline of the dominant kernel (rank 0's launches)
    roofline = None
    if recs:
        ms = np.array([r[1] for r in recs])
        by = np.array([r[2] for r in recs])
        fl = np.array([r[3] for r in recs])
        achieved = float(by.mean() / (ms.mean() * 1e-3) / 1e9)
        layers = sorted({r[0].split("/", 1)[1] for r in recs})
        # HBM bytes per launch from the PMC passes of this same command (separate FETCH_SIZE / WRITE_SIZE passes,
        # gfx950 correction applied: tools/pmc_traffic.py) — rocprofv3 cannot wrap itself, so the committed summary
        # of the latest passes is read back; null when the kernel is not in it
        traffic = None
        try:
            with open(os.path.join(REPO, "profiles", "traffic_latest.json")) as f:
                traffic = json.load(f).get(dominant, {}).get("traffic_bytes_per_launch")
        except Exception:
            traffic = None
        roofline = {"bound": "hbm", "kernel": dominant, "layers": layers,
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "batches_in_flight": args.streams,
                    "launches": int(len(ms)), "avg_launch_us": round(float(ms.mean()) * 1e3, 2),
                    "algorithmic_bytes_per_launch": float(by.mean()),
                    "tflops": round(float(fl.mean() / (ms.mean() * 1e-3) / 1e12), 2)}
        if excl:
            ems = np.array([r[1] for r in excl]); eby = np.array([r[2] for r in excl])
            ea = float(eby.mean() / (ems.mean() * 1e-3) / 1e9)
            roofline["exclusive"] = {"note": "same kernel, one batch in flight, timed after the timed region; this is "
                                             "what rocprofv3 --kernel-trace (which serialises streams) reports",
                                     "avg_launch_us": round(float(ems.mean()) * 1e3, 2), "achieved": round(ea, 1),
                                     "frac": round(ea / HBM_PEAK_GBS, 4)}

    # ---------------- optional per-layer table (outside the timed region)
    if args.layer_table and rank == 0:
        ctx.profile_enable(1)
        for _ in range(5):
            step()
        table = {}
        for name, t, b, f in ctx.profile_fetch():
            e = table.setdefault(name, {"ms": [], "bytes": b, "flops": f})
            e["ms"].append(t)
        rows = [{"kernel": k, "avg_us": float(np.mean(v["ms"])) * 1e3, "alg_bytes": v["bytes"], "flops": v["flops"],
                 "alg_GBps": v["bytes"] / (np.mean(v["ms"]) * 1e-3) / 1e9,
                 "TFLOPs": v["flops"] / (np.mean(v["ms"]) * 1e-3) / 1e12} for k, v in table.items()]
        ctx.profile_enable(0)
        with open(args.layer_table, "w") as f:
            json.dump({"levels": n_levels, "batch": args.batch, "rows": rows}, f, indent=1)

    # ---------------- CPU baseline: the oracle ("port") on a bounded sample of the same workload
    # oracle/egonn_cpu.c = C/OpenMP restatement of the reference path (one scan per forward, like the reference's
    # evaluator), built with -march=native on this box and run on all host cores; whole passes over the benchmark's
    # scans until >= --cpu-seconds of CPU work.  (numpy oracle as the fallback if gcc is unavailable.)
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            from oracle import egonn_cpu
            co = egonn_cpu.CpuOracle(sd, args.voxel, native=True)
            co.compute_embedding(scans[0][:5000], 128)                   # warm-up (thread pool)
            cores = os.cpu_count() or 1
            # scans in flight x OpenMP threads per scan: pick the best split of the host cores on a short trial
            # (128 threads on one 26k-voxel scan scale badly; the reference's DataLoader workers are processes too)
            splits = sorted({(w, max(1, cores // w)) for w in (1, 4, 16) if w <= max(1, cores)})
            trials = {sp: co.throughput(scans, sp[0], sp[1], 1.0)[0] for sp in splits}
            best = max(trials, key=trials.get)
            rate, done, secs = co.throughput(scans, best[0], best[1], args.cpu_seconds)
            cpu_baseline = {"value": round(rate, 3), "unit": "scans/s", "cores": int(best[0] * best[1]), "kind": "port",
                            "sample": f"{done} scans ({done // len(scans)} passes over the {len(scans)} benchmark scans), "
                                      f"C/OpenMP restatement of the reference path (oracle/egonn_cpu.c: voxelise + forward "
                                      f"+ top-128, one scan per forward, -O3 -march=native), {best[0]} scans in flight x "
                                      f"{best[1]} OpenMP threads (best of {dict((f'{k[0]}x{k[1]}', round(v, 1)) for k, v in trials.items())} "
                                      f"scans/s on 1 s trials), {secs:.1f} s of CPU work"}
        except Exception as e:                                            # pragma: no cover
            from oracle import egonn_ref as ref
            oracle = ref.EgoNNOracle(sd, ref.CartesianQuantizer(args.voxel))
            k = min(args.cpu_scans, len(scans))
            ref.compute_embedding(oracle, scans[0][:5000], 128)
            c0 = time.perf_counter()
            for i in range(k):
                ref.compute_embedding(oracle, scans[i], 128)
            c1 = time.perf_counter()
            cpu_baseline = {"value": round(k / (c1 - c0), 3), "unit": "scans/s", "cores": os.cpu_count() or 1,
                            "kind": "port", "sample": f"{k} scans on the numpy oracle (C oracle unavailable: {e}), "
                                                      f"{c1 - c0:.1f} s of CPU work"}

    if rank == 0:
        total_scans = args.batch * world * args.steps
        line = {
            "metric": "LiDAR scans/sec (descriptor extraction), 50k-pt clouds @ 0.1m voxel",
            "value": round(total_scans / elapsed, 2),
            "unit": "scans/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.dtype == "f32" else "bf16 (MFMA operands; f32 accumulate and feature maps)",
            "data": "synthetic",
            "config": {"workload": ("configs[1]" if args.dtype == "f32" and args.batch == 16 else "configs[1] variant") +
                                   ": EgoNN (minkgl) inference, synthetic 50k-pt LiDAR-like clouds, "
                                   f"Cartesian 0.1 m voxels, batch {args.batch} per GPU, {args.dtype}; step = voxelise + "
                                   "forward + top-128 keypoints; random-init weights",
                       "batch_per_gpu": args.batch, "points_per_scan": args.points, "voxel_m": args.voxel,
                       "voxels_per_level": n_levels, "parallelism": f"scan-sharded x{world} (no collective)", "batches_in_flight": args.streams},
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
