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
    def extract_packed(self, points: torch.Tensor, offsets: List[int], model: MinkGL = None):
        """points: (sum n_i, 3) float32 already resident on the device; offsets: host list of B+1 scan bounds."""
        model = model or self.model
        ctx = model.context()
        q = self.quantizer
        ctx.voxelize(points, offsets, q.mode, q.step)
        n0 = ctx.level_count(0)
        feats = self._ones(n0, ctx.device)
        y = model._forward_on_plan(ctx, feats)
        d, k, s = model._last_local
        sel_kp, sel_desc, rows, cnt = ctx.select_keypoints(s, k, d, self.n_k)
        return {'global': y['global'], 'keypoints': sel_kp, 'descriptors': sel_desc, 'count': cnt, 'rows': rows}

    def _ones(self, n, dev):
        buf = getattr(self, '_ones_buf', None)
        if buf is None or buf.shape[0] < n or buf.device != dev:
            buf = torch.ones((max(n, 1) * 5 // 4 + 1024, 1), dtype=torch.float32, device=dev)
            self._ones_buf = buf
        return buf[:n]
