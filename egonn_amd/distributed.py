"""Multi-GPU descriptor extraction: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI
on ROCm; "gloo" in the CPU tests).

The path shards naturally: every scan is an independent unit in eval mode (BatchNorm uses
running statistics, ECA / GeM reduce per sample), so the scan stream is partitioned contiguously over the
ranks and the forward needs NO collective.  The only exchange step of the database build (BASELINE.json
configs[4]) is one all-gather of the per-rank global descriptors at the end — 256 floats per scan, 20.5 MB for
20 000 scans — after which every rank (or just rank 0) holds the (N, 256) matrix that the reference's evaluator
feeds to its kNN (eval/evaluate.py:168-184).  Keypoints / local descriptors (67 KB per scan) stay rank-local.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import collections

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced partition: the first (n_items % world) ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


# collectives issued by this process through the helpers below, by kind ("all_reduce", "all_gather"): the sharded-step tests
# pin the communication pattern of a training step with it (tests/test_gpu_train.py); never read by the product path
COLLECTIVES = collections.Counter()


def _staged(t: torch.Tensor) -> bool:
    """gloo has no device collectives on this build: stage HIP tensors through the host (tests only — on the GPU
    box the backend is "nccl" = RCCL and tensors stay on the device)."""
    return t.is_cuda and dist.get_backend() == "gloo"


def all_reduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    """in-place SUM all-reduce (RCCL; host-staged under gloo)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t
    COLLECTIVES["all_reduce"] += 1
    if _staged(t):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def _all_gather_list(t: torch.Tensor, world: int) -> List[torch.Tensor]:
    COLLECTIVES["all_gather"] += 1
    if _staged(t):
        h = t.cpu()
        parts = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(parts, h)
        return [p.to(t.device) for p in parts]
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    return parts


def all_gather_rows(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """Concatenate the ranks' row blocks (contiguous `shard_bounds` partition of n_total rows) on every rank.
    One collective: blocks are padded to the largest shard so that a single all_gather_into_tensor
    (RCCL: one ring/tree all-gather over xGMI) moves everything."""
    rank, world = _world()
    if world == 1:
        assert local.shape[0] == n_total
        return local
    max_rows = (n_total + world - 1) // world
    pad = torch.zeros((max_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * max_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    COLLECTIVES["all_gather"] += 1
    if dist.get_backend() == "gloo":                      # gloo moves host memory: stage device tensors through the CPU
        hpad = pad.cpu()
        parts = [torch.empty_like(hpad) for _ in range(world)]
        dist.all_gather(parts, hpad)
        out = torch.cat(parts, dim=0).to(local.device)
    else:
        dist.all_gather_into_tensor(out, pad)
    chunks = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, r, world)
        chunks.append(out[r * max_rows: r * max_rows + (hi - lo)])
    return torch.cat(chunks, dim=0)


class DatabaseBuilder:
    """Streams a set of scans through the extractor, sharded over the ranks.

    extractor: object with `extract(list_of_scans) -> {'global': (B,256), 'keypoints': (B,n_k,3),
    'descriptors': (B,n_k,128), 'count': (B,)}` (egonn_amd.DescriptorExtractor, or a stand-in in CPU tests).
    load_scan: callable index -> (n,3) float32 tensor/array (dataset access is the caller's business).
    """

    def __init__(self, extractor, batch_size: int = 16):
        self.extractor = extractor
        self.batch_size = batch_size

    def build(self, load_scan: Callable[[int], torch.Tensor], n_scans: int, keep_local: bool = True) -> Dict:
        rank, world = _world()
        lo, hi = shard_bounds(n_scans, rank, world)
        globals_, kps, descs, counts = [], [], [], []
        for start in range(lo, hi, self.batch_size):
            idx = range(start, min(start + self.batch_size, hi))
            out = self.extractor.extract([load_scan(i) for i in idx])
            globals_.append(out["global"])
            if keep_local:
                kps.append(out["keypoints"])
                descs.append(out["descriptors"])
                counts.append(out["count"])
        # a rank whose shard is empty (fewer scans than ranks) still joins the collective: on the extractor's device, with
        # the model's descriptor size
        model = getattr(self.extractor, "model", None)
        dim = globals_[0].shape[1] if globals_ else int(getattr(model, "global_descriptor_size", 256))
        if globals_:
            dev = globals_[0].device
        elif model is not None and hasattr(model, "context"):
            dev = model.context().device
        else:
            dev = torch.device(getattr(self.extractor, "device", "cpu"))
        local = torch.cat(globals_, dim=0) if globals_ else torch.zeros((0, dim), device=dev)
        result = {"global": all_gather_rows(local, n_scans), "range": (lo, hi)}
        if keep_local and kps:
            result.update(keypoints=torch.cat(kps), descriptors=torch.cat(descs), count=torch.cat(counts))
        return result


def build_database_streaming(streamer, sources: Sequence, keep_local: bool = False) -> Dict:
    """BASELINE configs[4] with the streaming pipeline (egonn_amd/stream.py): `sources` = ALL raw scans of the database (host
    arrays (n,4|3) f32 or `.bin` paths; every rank holds the list, only its contiguous shard is read), `streamer` a calibrated
    StreamingExtractor.  Each rank streams its shard (S batches in flight, no host synchronisation between file and
    descriptor), then ONE all-gather of the (n_local, 256) descriptors (RCCL over xGMI).  Returns the (N,256) matrix on the
    extractor's device (+ rank-local keypoints / descriptors on the host with keep_local)."""
    rank, world = _world()
    n_scans = len(sources)
    lo, hi = shard_bounds(n_scans, rank, world)
    B = streamer.batch_size
    # keep_local is part of what a slot's captured graph copies back: it is fixed by the FIRST run of a streamer (the slots
    # are built then) and a later call with another value is refused rather than silently changing the streamer
    if streamer.slots and streamer.keep_local != keep_local:
        raise ValueError("build_database_streaming: the streamer's slots were built with a different keep_local")
    streamer.keep_local = keep_local
    batches = ([sources[i] for i in range(s0, min(s0 + B, hi))] for s0 in range(lo, hi, B))
    globals_, kps, descs, counts = [], [], [], []
    for out in streamer.run(batches):
        globals_.append(out["global"])
        counts.append(out["count"])
        if keep_local:
            kps.append(out["keypoints"])
            descs.append(out["descriptors"])
    model = streamer.extractor.model
    dev = model.context().device
    dim = int(model.global_descriptor_size)
    local = torch.cat(globals_, dim=0).to(dev) if globals_ else torch.zeros((0, dim), device=dev)
    result = {"global": all_gather_rows(local, n_scans), "range": (lo, hi), "fallbacks": streamer.fallbacks}
    if counts:
        result["count"] = torch.cat(counts)
    if keep_local and kps:
        result.update(keypoints=torch.cat(kps), descriptors=torch.cat(descs))
    return result


class _AllGatherEmbeddings(torch.autograd.Function):
    """Forward: all-gather of the per-rank (b_local, D) global descriptors into the (B, D) matrix the batch-hard
    miner needs (training/trainer.py:163-165 computes the loss on the whole batch).  Backward: every rank evaluates
    the SAME loss on the SAME gathered matrix, so dLoss/dE is identical everywhere and each rank keeps the rows it
    produced — no second collective; the later SUM all-reduce of parameter gradients then yields the full-batch
    gradient."""

    @staticmethod
    def forward(ctx, local, sizes):
        rank, world = _world()
        ctx.n_local = local.shape[0]
        if world == 1:
            ctx.lo = 0
            return local.clone()
        if sizes is None:                                 # shard sizes unknown: one extra (tiny) collective + host syncs
            counts = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
            sizes = [int(c.item()) for c in _all_gather_list(counts, world)]
        sizes = [int(v) for v in sizes]
        if len(sizes) != world or sizes[rank] != local.shape[0]:
            raise ValueError(f"all_gather_embeddings: shard sizes {sizes} do not match rank {rank}'s {local.shape[0]} rows")
        ctx.lo = sum(sizes[:rank])
        mx = max(sizes)
        pad = local
        if local.shape[0] != mx:
            pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            pad[: local.shape[0]] = local
        if _staged(pad):
            parts = _all_gather_list(pad, world)
        else:                                             # ONE collective: RCCL all-gather of the padded blocks
            out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
            COLLECTIVES["all_gather"] += 1
            dist.all_gather_into_tensor(out, pad.contiguous())
            if all(v == mx for v in sizes):
                return out
            parts = [out[r * mx:(r + 1) * mx] for r in range(world)]
        return torch.cat([p[:v] for p, v in zip(parts, sizes)], dim=0)

    @staticmethod
    def backward(ctx, grad):
        return grad[ctx.lo: ctx.lo + ctx.n_local].contiguous(), None


def all_gather_embeddings(local: torch.Tensor, sizes: Optional[Sequence[int]] = None) -> torch.Tensor:
    """(b_local, D) -> (sum b_local, D) on every rank, differentiable (see _AllGatherEmbeddings).
    sizes: rows of every rank's shard, known to all ranks from the sampler (e.g. `shard_bounds`): the exchange is then
    exactly one all-gather with no host synchronisation; without it the sizes are exchanged first."""
    return _AllGatherEmbeddings.apply(local, None if sizes is None else tuple(int(v) for v in sizes))
