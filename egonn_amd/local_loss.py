"""Local-head training losses on the device — reference models/loss_utils.py:11-139 (`KeypointLoss`,
`CorrespondenceLoss`) and their per-pair driver `KeypointCorrLoss` (models/loss.py:32-92).

Same class names, constructor arguments, call signatures, returned `(loss, metrics)` and metric keys as the reference, so
`training/trainer.py:186-190` can call them unchanged.  What moves to libegonn_hip (include/egonn_hip.h):
  * the searches the reference does on dense `torch.cdist` matrices — nearest keypoint of the other scan in both
    directions, nearest cloud point of every keypoint (keypoints x 50 k points per scan) — `egonn_nn_search` /
    `egonn_matrix_min`: indices only, no matrix is materialised by the driver;
  * the (kp1 x kp2) descriptor-similarity matrix (`egonn_dense`), its row-wise softmax cross-entropy and d loss / d logits
    (`egonn_softmax_cross_entropy`) and both descriptor gradients (`egonn_dense`, `egonn_dense_backward_weight`).
The differentiable tail that touches only (n,3) / (n,1) tensors (distance of a keypoint to its selected partner, the
log-sigma terms, the means) is ordinary autograd on device tensors.  There is no CPU path: inputs must live on the GPU.
Golden vectors: tests/golden/local_losses_*.npz, produced by importing the reference module itself
(tests/golden/make_golden_losses.py).
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np
import torch

from . import _lib

EPS = 1e-5


def _dev(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("egonn_amd local losses run on the HIP device only (no CPU fallback)")
    return t


def apply_transform(pc: torch.Tensor, m: torch.Tensor) -> torch.Tensor:
    """misc/poses.py:68-76 (3-D case): pc @ m[:3,:3].T + m[:3,3]."""
    assert pc.dim() == 2 and pc.shape[1] == 3 and tuple(m.shape) == (4, 4)
    m = m.to(device=pc.device, dtype=pc.dtype)
    return pc @ m[:3, :3].transpose(1, 0) + m[:3, -1]


def nn_search(a: torch.Tensor, b: torch.Tensor):
    """index (int64) and distance of the nearest row of b (m,3) for every row of a (n,3): torch.min(torch.cdist(a, b), 1)."""
    a = _dev(a).detach().float().contiguous()
    b = _dev(b).detach().float().contiguous()
    dist = torch.empty(a.shape[0], dtype=torch.float32, device=a.device)
    idx = torch.empty(a.shape[0], dtype=torch.int32, device=a.device)
    if a.shape[0]:
        with torch.cuda.device(a.device):
            _lib.check(_lib.load().egonn_nn_search(a.data_ptr(), a.shape[0], None, b.data_ptr(), b.shape[0], dist.data_ptr(),
                                                   idx.data_ptr(), _lib._stream()))
    return dist, idx.long()


def matrix_min(d: torch.Tensor):
    """(row_min, row_idx, col_min, col_idx) of a dense (n,m) matrix = torch.min(d, 1) and torch.min(d, 0)."""
    d = _dev(d).detach().float().contiguous()
    n, m = d.shape
    rv = torch.empty(n, dtype=torch.float32, device=d.device)
    ri = torch.empty(n, dtype=torch.int32, device=d.device)
    cv = torch.empty(m, dtype=torch.float32, device=d.device)
    ci = torch.empty(m, dtype=torch.int32, device=d.device)
    with torch.cuda.device(d.device):
        _lib.check(_lib.load().egonn_matrix_min(d.data_ptr(), n, m, rv.data_ptr(), ri.data_ptr(), cv.data_ptr(), ci.data_ptr(),
                                                _lib._stream()))
    return rv, ri.long(), cv, ci.long()


class _RowNorm(torch.autograd.Function):
    """||x_i|| per row with the zero-distance gradient torch.cdist uses (0, not 0/0)."""

    @staticmethod
    def forward(ctx, x):
        d = x.pow(2).sum(dim=1).sqrt()
        ctx.save_for_backward(x, d)
        return d

    @staticmethod
    def backward(ctx, g):
        x, d = ctx.saved_tensors
        return torch.where(d.unsqueeze(1) > 0, x / d.clamp_min(1e-30).unsqueeze(1), torch.zeros_like(x)) * g.unsqueeze(1)


class _SimilarityCE(torch.autograd.Function):
    """sum over the kept rows of CrossEntropy(scale * desc1 @ desc2.T, target); rows with target < 0 are ignored."""

    @staticmethod
    def forward(ctx, desc1, desc2, target, scale):
        lib = _lib.load()
        d1 = _dev(desc1).detach().float().contiguous()
        d2 = _dev(desc2).detach().float().contiguous()
        n1, c = d1.shape
        n2 = d2.shape[0]
        dev = d1.device
        tg = target.to(device=dev, dtype=torch.int32).contiguous()
        logits = torch.empty((n1, n2), dtype=torch.float32, device=dev)
        rows = torch.empty(n1, dtype=torch.float32, device=dev)
        amax = torch.empty(n1, dtype=torch.int32, device=dev)
        dlog = torch.empty((n1, n2), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            # similarity = desc1 @ desc2.T : desc2 is an (out, in) "Linear weight"
            _lib.check(lib.egonn_dense(d1.data_ptr(), n1, c, d2.data_ptr(), 1, None, n2, 0, logits.data_ptr(), _lib._stream()))
            logits.mul_(scale)
            _lib.check(lib.egonn_softmax_cross_entropy(logits.data_ptr(), n1, n2, tg.data_ptr(), rows.data_ptr(), amax.data_ptr(),
                                                       dlog.data_ptr(), _lib._stream()))
        ctx.save_for_backward(d1, d2, dlog)
        ctx.scale = scale
        ctx.mark_non_differentiable(logits, amax)
        return rows.sum(), logits, amax.long()

    @staticmethod
    def backward(ctx, g, _gl, _ga):
        d1, d2, dlog = ctx.saved_tensors
        lib = _lib.load()
        dev = d1.device
        n1, c = d1.shape
        n2 = d2.shape[0]
        ds = (dlog * (g * ctx.scale)).contiguous()                   # d loss / d (desc1 @ desc2.T)
        g1 = torch.empty((n1, c), dtype=torch.float32, device=dev)
        g2 = torch.empty((n2, c), dtype=torch.float32, device=dev)
        scratch = torch.empty(max(64 * n2 * c, 1 << 20), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            # grad_desc1 = dS @ desc2 : desc2 as a (cin = n2, cout = c) kernel
            _lib.check(lib.egonn_dense(ds.data_ptr(), n1, n2, d2.data_ptr(), 0, None, c, 0, g1.data_ptr(), _lib._stream()))
            # grad_desc2 = dS.T @ desc1 : a^T b over the n1 rows
            _lib.check(lib.egonn_dense_backward_weight(ds.data_ptr(), n2, d1.data_ptr(), c, n1, g2.data_ptr(),
                                                       scratch.data_ptr(), scratch.numel(), _lib._stream()))
        return g1, g2, None, None


class KeypointLoss:
    """reference models/loss_utils.py:11-95 (probabilistic chamfer loss between the regressed keypoints of two scans +
    point-to-point loss to the clouds)."""

    def __init__(self, gamma_chamfer=1., gamma_p2p=1., prob_chamfer_loss=True, p2p_loss=True, repeatability_dist_th=0.5):
        self.gamma_chamfer = gamma_chamfer
        self.gamma_p2p = gamma_p2p
        self.prob_chamfer_loss = prob_chamfer_loss
        self.p2p_loss = p2p_loss
        self.repeatability_dist_th = repeatability_dist_th

    def __call__(self, pc1, kp1, sigma1, pc2, kp2, sigma2, dist_kp1_trans_kp2):
        assert pc1.shape[1] == 3 and pc2.shape[1] == 3 and kp1.shape[1] == 3 and kp2.shape[1] == 3
        assert sigma1.shape[1] == 1 and sigma2.shape[1] == 1
        assert kp1.shape[0] == sigma1.shape[0] and kp2.shape[0] == sigma2.shape[0]
        assert dist_kp1_trans_kp2.shape[0] == kp1.shape[0] and dist_kp1_trans_kp2.shape[1] == kp2.shape[0]
        _, ndx1, _, ndx2 = matrix_min(dist_kp1_trans_kp2)
        min_dist1 = dist_kp1_trans_kp2.gather(1, ndx1.unsqueeze(1)).squeeze(1)       # autograd flows into the matrix
        min_dist2 = dist_kp1_trans_kp2.gather(0, ndx2.unsqueeze(0)).squeeze(0)
        return self._finish(pc1, kp1, sigma1, pc2, kp2, sigma2, min_dist1, ndx1, min_dist2, ndx2)

    def _finish(self, pc1, kp1, sigma1, pc2, kp2, sigma2, min_dist1, min_dist_ndx1, min_dist2, min_dist_ndx2):
        sigma1 = sigma1.squeeze(1)
        sigma2 = sigma2.squeeze(1)
        if self.prob_chamfer_loss:                                   # loss_utils.py:50-63
            sigma12 = (sigma1 + sigma2[min_dist_ndx1]) / 2
            loss1 = (torch.log(sigma12) + min_dist1 / sigma12).mean()
            sigma21 = (sigma2 + sigma1[min_dist_ndx2]) / 2
            loss2 = (torch.log(sigma21) + min_dist2 / sigma21).mean()
        else:
            loss1, loss2 = min_dist1.mean(), min_dist2.mean()
        metrics = {}
        metrics['repeatability'] = torch.mean((min_dist1 <= self.repeatability_dist_th).float()).item()
        metrics['chamfer_pure'] = 0.5 * (min_dist1.detach().mean() + min_dist2.detach().mean()).item()
        if self.prob_chamfer_loss:
            w12 = (1.0 / sigma12.detach()) / (1.0 / sigma12.detach()).mean()
            w21 = (1.0 / sigma21.detach()) / (1.0 / sigma21.detach()).mean()
            metrics['chamfer_weighted'] = (0.5 * (w12 * min_dist1.detach()).mean() + 0.5 * (w21 * min_dist2.detach()).mean()).item()
        # (like the reference, 'mean_sigma' needs prob_chamfer_loss: loss_utils.py:76 reads sigma12 unconditionally)
        metrics['mean_sigma'] = 0.5 * (sigma12.detach().mean() + sigma21.detach().mean()).item()
        loss = self.gamma_chamfer * 0.5 * (loss1 + loss2)
        metrics['loss_chamfer'] = loss.item()
        if self.p2p_loss:                                            # loss_utils.py:80-91, without the (n_kp, n_points) matrix
            _, i1 = nn_search(kp1, pc1)
            _, i2 = nn_search(kp2, pc2)
            d1 = _RowNorm.apply(kp1 - pc1[i1])
            d2 = _RowNorm.apply(kp2 - pc2[i2])
            loss_p2p = 0.5 * (d1.mean() + d2.mean())
            metrics['loss_p2p'] = loss_p2p.item()
            loss = loss + self.gamma_p2p * loss_p2p
        metrics['keypoint_loss'] = loss.item()
        return loss, metrics


class CorrespondenceLoss(torch.nn.Module):
    """reference models/loss_utils.py:98-139 (cross-entropy over the descriptor-similarity matrix, classes = the nearest
    keypoint of the other scan within dist_th)."""

    def __init__(self, beta, dist_th=0.5):
        super().__init__()
        self.beta = beta
        self.dist_th = dist_th

    def forward(self, desc1, desc2, dist_kp1_trans_kp2):
        assert dist_kp1_trans_kp2.shape[0] == desc1.shape[0] and dist_kp1_trans_kp2.shape[1] == desc2.shape[0]
        min_dist1, min_dist_ndx1, _, _ = matrix_min(dist_kp1_trans_kp2)
        return self._finish(desc1, desc2, min_dist1, min_dist_ndx1)

    def _finish(self, desc1, desc2, min_dist1, min_dist_ndx1):
        mask = min_dist1.detach() <= self.dist_th
        target = torch.where(mask, min_dist_ndx1, torch.full_like(min_dist_ndx1, -1))
        total, sim, amax = _SimilarityCE.apply(desc1, desc2, target, math.exp(self.beta))
        matching_keypoints = torch.sum(mask).float().item()
        loss = total / matching_keypoints if matching_keypoints > 0 else total * float('nan')   # CrossEntropyLoss(mean) of no rows
        if matching_keypoints > 0:
            tgt = min_dist_ndx1[mask]
            matching_descriptors = torch.sum(amax[mask] == tgt).float().item()
            pos_similarity = torch.mean(amax[mask].float()).item()          # (sic) the reference averages the arg-max index
            neg_mat = sim[mask].clone()
            neg_mat[:, tgt] = 0.
            neg_similarity = torch.mean(torch.max(neg_mat, 1)[0]).float().item()
        else:
            matching_descriptors = pos_similarity = neg_similarity = 0.
        metrics = {'correspondence_loss': loss.item(), 'matching_keypoints': matching_keypoints,
                   'matching_descriptors': matching_descriptors, 'pos_similarity': pos_similarity,
                   'neg_similarity': neg_similarity}
        return loss, metrics


def metrics_mean(l: List[Dict]) -> Dict:
    """reference models/loss_utils.py:142-154"""
    metrics = {}
    for e in l:
        for k, v in e.items():
            metrics.setdefault(k, []).append(v)
    return {k: np.mean(np.array(v)) for k, v in metrics.items()}


class KeypointCorrLoss:
    """reference models/loss.py:32-92: keypoint loss + correspondence loss per (anchor, positive) pair of scans, averaged
    over the batch.  The (kp1 x kp2) distance matrix of the reference is never built: the nearest partners come from
    `egonn_nn_search`, the distances to them are recomputed differentiably."""

    def __init__(self, gamma_c=1., gamma_k=1., gamma_chamfer=1., gamma_p2p=1., beta=1., dist_th=0.5):
        self.keypoint_loss = KeypointLoss(gamma_chamfer=gamma_chamfer, gamma_p2p=gamma_p2p, prob_chamfer_loss=True,
                                          p2p_loss=True, repeatability_dist_th=dist_th)
        self.correspondence_loss = CorrespondenceLoss(beta=beta, dist_th=dist_th)
        self.gamma_k = gamma_k
        self.gamma_c = gamma_c

    def __call__(self, clouds1, keypoints1, sigma1, descriptors1, clouds2, keypoints2, sigma2, descriptors2, M_gt, len_batch):
        assert clouds1.dim() == 2 and clouds2.dim() == 2
        assert len(keypoints1) == len(sigma1) == len(descriptors1) and len(keypoints2) == len(sigma2) == len(descriptors2)
        cum1 = np.cumsum([0] + [e[0] for e in len_batch])
        cum2 = np.cumsum([0] + [e[1] for e in len_batch])
        assert cum1[-1] == len(clouds1) and cum2[-1] == len(clouds2)
        batch_metrics, batch_loss = [], []
        for i, (kp1, s1, d1, kp2, s2, d2, Mi) in enumerate(zip(keypoints1, sigma1, descriptors1, keypoints2, sigma2,
                                                              descriptors2, M_gt)):
            pc1 = clouds1[cum1[i]:cum1[i + 1]]
            pc2 = clouds2[cum2[i]:cum2[i + 1]]
            kp1_trans = apply_transform(kp1, torch.as_tensor(Mi))
            _, ndx1 = nn_search(kp1_trans, kp2)                      # = torch.min(torch.cdist(kp1_trans, kp2), 1)
            _, ndx2 = nn_search(kp2, kp1_trans)                      # = torch.min(..., 0)
            min_dist1 = _RowNorm.apply(kp1_trans - kp2[ndx1])
            min_dist2 = _RowNorm.apply(kp1_trans[ndx2] - kp2)
            metrics = {'kp_per_cloud': 0.5 * (len(kp1) + len(kp2))}
            loss_k, m = self.keypoint_loss._finish(pc1, kp1, s1, pc2, kp2, s2, min_dist1, ndx1, min_dist2, ndx2)
            metrics.update(m)
            loss_c, m = self.correspondence_loss._finish(d1, d2, min_dist1, ndx1)
            metrics.update(m)
            loss = self.gamma_k * loss_k + self.gamma_c * loss_c
            metrics['loss'] = loss.item()
            batch_metrics.append(metrics)
            batch_loss.append(loss)
        return torch.stack(batch_loss).mean(), metrics_mean(batch_metrics)


def make_local_loss(loss_gammas=None) -> KeypointCorrLoss:
    """the `loc_loss_fn` of reference models/loss.py:12-29 (`make_losses`): loss_gammas = [gamma_chamfer, gamma_p2p, gamma_c,
    beta], default [1, 1, 1, 2]."""
    gamma_chamfer, gamma_p2p, gamma_c, beta = loss_gammas if loss_gammas is not None else [1., 1., 1., 2.]
    return KeypointCorrLoss(gamma_c=gamma_c, gamma_chamfer=gamma_chamfer, gamma_p2p=gamma_p2p, beta=beta)
