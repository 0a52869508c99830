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
        n = int(out['count'][0].item())
        global_embedding = out['global'].cpu().numpy()
        return global_embedding, out['keypoints'][0, :n].cpu(), out['descriptors'][0, :n].cpu()

    def get_keypoints_idxes(self, sigmas: torch.Tensor, n_k: int):
        """rows (into the per-sample local outputs of the LAST forward of sample 0) of the n_k lowest sigmas."""
        ctx = self.model.context()
        d, k, s = self.model._last_local
        _, _, rows, cnt = ctx.select_keypoints(s, k, d, n_k)
        off = ctx.level_batch_offsets(3)
        return (rows[0, :int(cnt[0].item())].long() - off[0])

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
        sel_kp, sel_desc, rows, cnt = ctx.select_keypoints(s, k, d, self.n_k)
        return {'global': y['global'], 'keypoints': sel_kp, 'descriptors': sel_desc, 'count': cnt, 'rows': rows}

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
