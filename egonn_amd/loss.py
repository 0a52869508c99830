"""Batch-hard triplet loss with masks on the device — reference models/loss.py:146-172 (and the miner :95-143).

    loss_fn = BatchHardTripletLossWithMasks(margin=0.2)
    loss, stats, hard_triplets = loss_fn(embeddings, positives_mask, negatives_mask)

`loss` is a 0-d tensor wired into autograd (its backward hands dLoss/dEmbeddings, computed by the same HIP call,
to whatever produced `embeddings` — e.g. `egonn_amd.distributed.all_gather_embeddings`); `stats` has the
reference's keys; `hard_triplets` = (a, p, n) index tensors.  Arithmetic: libegonn_hip (no torch fallback).
"""
from __future__ import annotations

import torch

from . import _lib


class _TripletLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, embeddings, pos_mask, neg_mask, margin):
        lib = _lib.load()
        e = embeddings.detach().contiguous().float()
        assert e.is_cuda and e.dim() == 2, "batch-hard triplet loss runs on the HIP device only"
        n, d = e.shape
        pm = pos_mask.to(device=e.device, dtype=torch.uint8).contiguous()
        nm = neg_mask.to(device=e.device, dtype=torch.uint8).contiguous()
        assert pm.shape == (n, n) and nm.shape == (n, n)
        stats = torch.empty(10, dtype=torch.float32, device=e.device)
        trip = torch.empty((n, 3), dtype=torch.int32, device=e.device)
        grad = torch.empty_like(e)
        scratch = torch.empty(lib.egonn_triplet_loss_scratch_floats(n), dtype=torch.float32, device=e.device)
        with torch.cuda.device(e.device):
            _lib.check(lib.egonn_triplet_loss(e.data_ptr(), n, d, pm.data_ptr(), nm.data_ptr(), float(margin),
                                              stats.data_ptr(), trip.data_ptr(), grad.data_ptr(), scratch.data_ptr(),
                                              _lib._stream()))
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(stats, trip)
        return stats[0].clone(), stats, trip

    @staticmethod
    def backward(ctx, g_loss, g_stats, g_trip):
        (grad,) = ctx.saved_tensors
        return grad * g_loss, None, None, None


class BatchHardTripletLossWithMasks:
    def __init__(self, margin: float):
        self.margin = margin

    def __call__(self, embeddings, positives_mask, negatives_mask):
        loss, st, trip = _TripletLossFn.apply(embeddings, positives_mask, negatives_mask, self.margin)
        s = st.tolist()                                    # the reference also syncs here (.item() calls)
        keep = trip[:, 0] >= 0
        hard_triplets = (trip[keep, 0].long(), trip[keep, 1].long(), trip[keep, 2].long())
        stats = {'loss': s[0], 'avg_embedding_norm': s[3], 'num_non_zero_triplets': int(s[2]),
                 'num_triplets': int(s[1]), 'mean_pos_pair_dist': s[4], 'mean_neg_pair_dist': s[7],
                 'max_pos_pair_dist': s[5], 'max_neg_pair_dist': s[8], 'min_pos_pair_dist': s[6],
                 'min_neg_pair_dist': s[9]}
        return loss, stats, hard_triplets


def make_losses(margin: float = 0.2):
    """reference models/loss.py:12-29 for loss = BatchHardTripletMarginLoss (config/config_egonn.txt:20-22)."""
    return BatchHardTripletLossWithMasks(margin)
