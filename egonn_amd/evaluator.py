"""Descriptor extraction front end — the counterpart of the reference's
`MinkLocGLEvaluator.compute_embedding` / `get_keypoints_idxes` (eval/evaluate.py:327-361).

Two entry points:
  * `compute_embedding(pc, model)` — the reference's per-scan contract: global (1,256) numpy,
    keypoints (n_k,3) and descriptors (n_k,128) CPU tensors in sigma-ascending order.
  * `extract(scans)` — the batched on-device pipeline used by bench.py and the multi-GPU database build:
    voxelise -> forward -> top-k with a single host synchronisation (the size query) per batch.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch

from . import _lib
from .model import MinkGL


class DescriptorExtractor:
    def __init__(self, model: MinkGL, n_k: int = 128, ignore_keypoint_saliency: bool = False):
        self.model = model
        self.n_k = n_k
        self.quantizer = model.quantizer
        self.ignore_keypoint_saliency = ignore_keypoint_saliency

    # ------------------------------------------------------------------ reference-shaped API
    @torch.no_grad()
    def compute_embedding(self, pc, model: MinkGL = None):
        model = model or self.model
        pc = pc if isinstance(pc, torch.Tensor) else torch.as_tensor(np.asarray(pc), dtype=torch.float32)
        out = self.extract([pc], model=model)
        ctx = model.context()
        try:                                           # (the host waits for the results below anyway)
            ctx.plan_status()
        except _lib.Fp16RangeError:
            # an activation left the range of the fp16 operand parts (models/minkgl.py:105 is fp32 arithmetic): this scan again on
            # the exact fp32 kernels
            ctx.set_exact_fp32(True)
            try:
                out = self.extract([pc], model=model)
                ctx.plan_status()
            finally:
                ctx.set_exact_fp32(False)
        n = int(out['count'][0].item())
        global_embedding = out['global'].cpu().numpy()
        return global_embedding, out['keypoints'][0, :n].cpu(), out['descriptors'][0, :n].cpu()

    def get_keypoints_idxes(self, sigmas: torch.Tensor, n_k: int):
        """eval/evaluate.py:352-361: indices (into `sigmas`, the (n,1) saliency of ONE scan) of the n_k keypoints with the
        lowest sigma, ascending; with `ignore_keypoint_saliency` n_k random indices instead (:354-356)."""
        n = sigmas.shape[0]
        if self.ignore_keypoint_saliency:
            return torch.randperm(n)[:min(n_k, n)]
        ctx = self.model.context()
        dev = ctx.device
        sg = sigmas.detach().to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        # a one-scan selection on the given saliencies: the library's select kernel with explicit offsets
        rows = torch.empty((1, n_k), dtype=torch.int32, device=dev)
        cnt = torch.empty((1,), dtype=torch.int32, device=dev)
        boff = torch.tensor([0, n], dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(ctx.lib.egonn_topk_rows(sg.data_ptr(), boff.data_ptr(), 1, n_k, rows.data_ptr(), cnt.data_ptr(),
                                               _lib._stream()))
        return rows[0, :int(cnt[0].item())].long().cpu()

    # ------------------------------------------------------------------ batched device pipeline
    @torch.no_grad()
    def extract(self, scans: Sequence[torch.Tensor], model: MinkGL = None) -> Dict[str, torch.Tensor]:
        """scans: list of (n_i,3) float32 tensors (any device).  Returns device tensors:
        global (B,256), keypoints (B,n_k,3), descriptors (B,n_k,128), count (B,), rows (B,n_k)."""
        model = model or self.model
        ctx = model.context()
        dev = ctx.device
        pts = [torch.as_tensor(s, dtype=torch.float32).to(dev) for s in scans]
        offsets = [0]
        for p in pts:
            offsets.append(offsets[-1] + p.shape[0])
        allpts = pts[0].contiguous() if len(pts) == 1 else torch.cat(pts, dim=0)
        return self.extract_packed(allpts, offsets, model)

    @torch.no_grad()
    def extract_packed(self, points: torch.Tensor, offsets: List[int], model: MinkGL = None, slot: int = 0):
        """points: (sum n_i, 3) float32 already resident on the device; offsets: host list of B+1 scan bounds.
        `slot` selects the egonn_ctx (plan + workspace): batches on different slots may be in flight at once."""
        model = model or self.model
        ctx = model.context(slot)
        q = self.quantizer
        ctx.voxelize(points, offsets, q.mode, q.step)
        y = model._forward_on_plan(ctx, None)          # unit features (eval/evaluate.py:334) -> occupancy-only conv0
        d, k, s = model._last_local
        if self.ignore_keypoint_saliency:              # eval/evaluate.py:354-356: n_k random keypoints per scan
            s = torch.rand_like(s)
        sel_kp, sel_desc, rows, cnt = ctx.select_keypoints(s, k, d, self.n_k)
        return {'global': y['global'], 'keypoints': sel_kp, 'descriptors': sel_desc, 'count': cnt, 'rows': rows}

    # ------------------------------------------------------------------ captured (hipGraph) pipeline
    @torch.no_grad()
    def calibrate(self, points: torch.Tensor, offsets: List[int], margin: float = 1.3):
        """Level capacities for `GraphExtractor` from one representative batch (eager run): margin x the observed rows."""
        ctx = self.model.context(0)
        q = self.quantizer
        ctx.voxelize(points, offsets, q.mode, q.step)
        return [int(ctx.level_count(l) * margin) + 1024 for l in range(8)]

    def graph(self, batch_size: int, max_points: int, level_capacity=None, slot: int = 0, stream=None):
        return GraphExtractor(self, batch_size, max_points, level_capacity, slot, stream)

    @torch.no_grad()
    def extract_stream(self, batches, n_streams: int = 2, model: MinkGL = None):
        """Throughput mode: `batches` yields (points, offsets) with the points resident on the device; batch i runs
        on HIP stream i % n_streams with its own egonn_ctx, so the latency-bound tail of one batch (small levels,
        heads, top-k: few workgroups) overlaps the bandwidth/MFMA-bound head of the next.  The per-batch size query
        only blocks the host on that batch's stream.  Yields the per-batch result dicts in order; results are valid
        after `torch.cuda.synchronize()` (or a sync on the batch's stream)."""
        model = model or self.model
        dev = model.context(0).device
        model._sync_weights()
        if getattr(self, '_streams', None) is None or len(self._streams) != n_streams:
            self._streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
        torch.cuda.current_stream(dev).synchronize()
        for i, (points, offsets) in enumerate(batches):
            slot = i % n_streams
            with torch.cuda.stream(self._streams[slot]):
                yield self.extract_packed(points, offsets, model, slot=slot)

    def _ones(self, n, dev):
        # shared read-only buffer (filled once on the default stream before any side stream uses it)
        buf = getattr(self, '_ones_buf', None)
        if buf is None or buf.shape[0] < n or buf.device != dev:
            buf = torch.ones((max(n, 1) * 5 // 4 + 1024, 1), dtype=torch.float32, device=dev)
            torch.cuda.synchronize(dev)
            self._ones_buf = buf
        return buf[:n]


class GraphExtractor:
    """BASELINE configs[2]: voxelise -> forward -> top-n_k captured ONCE into a hipGraph and replayed per batch.

    The context is reserved (egonn_ctx_reserve) for `max_points` input rows, `batch_size` scans and the given level
    capacities; the level sizes of every batch stay in device memory, so the captured launches are valid for any batch
    that fits.  `run(points, offsets)` copies the batch into the static input buffers, replays the graph (one host call
    instead of ~150 launches) and returns the static output tensors; `status()` synchronises and raises if a batch left
    the coordinate range or the reservation."""

    def __init__(self, extractor: DescriptorExtractor, batch_size: int, max_points: int, level_capacity=None, slot: int = 0,
                 stream=None):
        self.ex = extractor
        model = extractor.model
        self.model = model
        self.ctx = model.context(slot)
        self.B = int(batch_size)
        self.max_points = int(max_points)
        ctx, dev = self.ctx, self.ctx.device
        model._sync_weights()
        ctx.reserve(self.max_points, self.B, level_capacity)
        self.points = torch.zeros((self.max_points, 3), dtype=torch.float32, device=dev)
        self.offsets = torch.zeros((self.B + 1,), dtype=torch.int64, device=dev)
        self._host_off = torch.zeros((self.B + 1,), dtype=torch.int64).pin_memory()
        self._off_copied = None                      # event: the previous batch's offsets left the pinned buffer
        # `stream`: a caller-made stream (e.g. torch.cuda.ExternalStream around hipExtStreamCreateWithCUMask: one CU
        # partition per batch in flight); default: a fresh stream of this device
        self.stream = stream if stream is not None else torch.cuda.Stream(device=dev)
        self.graph = None
        self.out = None

    def _enqueue(self):
        ctx, model, q = self.ctx, self.model, self.ex.quantizer
        ctx.voxelize_device(self.points, self.offsets, self.B, q.mode, q.step)
        if self.out is None:
            dev = ctx.device
            cap3 = ctx.level_capacity(3)
            n_k = self.ex.n_k
            self.out = {
                'global': torch.zeros((self.B, model.global_descriptor_size), dtype=torch.float32, device=dev),
                'all_descriptors': torch.zeros((cap3, model.local_descriptor_size), dtype=torch.float32, device=dev),
                'all_keypoints': torch.zeros((cap3, 3), dtype=torch.float32, device=dev),
                'all_sigma': torch.zeros((cap3, 1), dtype=torch.float32, device=dev),
                'keypoints': torch.zeros((self.B, n_k, 3), dtype=torch.float32, device=dev),
                'descriptors': torch.zeros((self.B, n_k, model.local_descriptor_size), dtype=torch.float32, device=dev),
                'rows': torch.zeros((self.B, n_k), dtype=torch.int32, device=dev),
                'count': torch.zeros((self.B,), dtype=torch.int32, device=dev),
            }
        o = self.out
        model._forward_on_plan(ctx, None, outputs=(o['global'], o['all_descriptors'], o['all_keypoints'], o['all_sigma']))
        with torch.cuda.device(ctx.device):
            _lib.check(ctx.lib.egonn_select_keypoints(ctx.h, o['all_sigma'].data_ptr(), o['all_keypoints'].data_ptr(),
                                                      o['all_descriptors'].data_ptr(), self.ex.n_k,
                                                      o['keypoints'].data_ptr(), o['descriptors'].data_ptr(),
                                                      o['rows'].data_ptr(), o['count'].data_ptr(), _lib._stream()))

    def _load(self, points: torch.Tensor, offsets):
        n = int(offsets[-1])
        if n > self.max_points or len(offsets) != self.B + 1:
            raise ValueError(f"batch of {n} points / {len(offsets) - 1} scans does not fit the reservation "
                             f"({self.max_points} points, {self.B} scans)")
        if self._off_copied is not None:
            self._off_copied.synchronize()           # (waits for a 100-byte copy, not for the graph)
        # the batch may have been produced on the caller's current stream (a cat / slice on the device): order the copy
        # below (on self.stream) behind it, and keep the caller's tensor alive until the copy has read it
        cur = torch.cuda.current_stream(self.ctx.device) if not hasattr(self, "_caller_stream") else self._caller_stream
        if cur != self.stream:
            self.stream.wait_stream(cur)
            if points.is_cuda:
                points.record_stream(self.stream)
        self._host_off.copy_(torch.as_tensor(list(offsets), dtype=torch.int64))
        self.points[:n].copy_(points[:n], non_blocking=True)
        self.offsets.copy_(self._host_off, non_blocking=True)
        self._off_copied = torch.cuda.Event()
        self._off_copied.record(self.stream)

    @torch.no_grad()
    def run(self, points: torch.Tensor, offsets):
        """points (n,3) f32 on the device, offsets: B+1 host ints.  Returns the static output dict (valid after a sync of
        `self.stream`; overwritten by the next run)."""
        ctx = self.ctx
        self._caller_stream = torch.cuda.current_stream(ctx.device)      # the stream the caller built `points` on
        with torch.cuda.stream(self.stream):
            self._load(points, offsets)
            if self.graph is None:
                self._enqueue()                       # eager once: grows every arena to its final size
                self.stream.synchronize()
                ctx.plan_status()
                g = _lib._P()
                _lib.check(ctx.lib.egonn_graph_begin(self.stream.cuda_stream))
                try:
                    self._enqueue()
                finally:
                    rc = ctx.lib.egonn_graph_end(self.stream.cuda_stream, _lib.C.byref(g))
                _lib.check(rc)
                self.graph = g
            _lib.check(ctx.lib.egonn_graph_launch(self.graph, self.stream.cuda_stream))
        return self.out

    def replay(self):
        """Replay the captured step on the batch that is already in the static input buffers (no copies)."""
        _lib.check(self.ctx.lib.egonn_graph_launch(self.graph, self.stream.cuda_stream))
        return self.out

    def status(self):
        """[SYNC] wait for the last run and raise if its batch was out of range / did not fit."""
        with torch.cuda.stream(self.stream):
            self.ctx.plan_status()

    def __del__(self):
        try:
            if getattr(self, 'graph', None):
                self.ctx.lib.egonn_graph_destroy(self.graph)
                self.graph = None
        except Exception:
            pass
