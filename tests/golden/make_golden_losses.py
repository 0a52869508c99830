"""Golden vectors of the local-head training losses.  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).

These are the one place where REAL reference outputs exist for this path: the reference's
`models/loss_utils.py` and `misc/poses.py` import and run here as they are (numpy + torch only).  The script imports them,
feeds seeded inputs to `KeypointLoss` (loss_utils.py:11-95) and `CorrespondenceLoss` (:98-139) exactly the way the
reference's driver does (`models/loss.py:71-86`: apply_transform -> torch.cdist -> the two losses -> gamma-weighted sum),
back-propagates, and stores inputs, losses, metrics and all six input gradients as plain arrays.  `models/loss.py` itself
cannot be imported (it needs pytorch_metric_learning), so the few driver lines are re-expressed here, in this script's own
words, on top of the imported reference classes.  No reference source text is stored.

    python tests/golden/make_golden_losses.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def unit(x):
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def make_pair(rng, n1, n2, m1, m2, overlap, noise):
    """two clouds of a scene seen from two poses + regressed keypoints / saliencies / descriptors of both"""
    ang = rng.uniform(-0.6, 0.6)
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    t = rng.uniform(-3, 3, 3) * np.array([1, 1, 0.1])
    M = np.eye(4)
    M[:3, :3], M[:3, 3] = R, t                      # cloud-1 frame -> cloud-2 frame
    pc1 = rng.uniform(-30, 30, (m1, 3)) * np.array([1, 1, 0.15])
    pc2 = np.concatenate([(pc1[: m2 // 2] @ R.T + t), rng.uniform(-30, 30, (m2 - m2 // 2, 3)) * np.array([1, 1, 0.15])])
    kp1 = pc1[rng.choice(m1, n1, replace=False)] + rng.normal(0, noise, (n1, 3))
    n_shared = int(overlap * min(n1, n2))
    kp2 = np.concatenate([kp1[:n_shared] @ R.T + t + rng.normal(0, noise, (n_shared, 3)),
                          pc2[rng.choice(m2, n2 - n_shared, replace=False)] + rng.normal(0, noise, (n2 - n_shared, 3))])
    d1 = unit(rng.standard_normal((n1, 128)))
    d2 = unit(np.concatenate([d1[:n_shared] + 0.4 * rng.standard_normal((n_shared, 128)),
                              rng.standard_normal((n2 - n_shared, 128))]))
    s1 = rng.uniform(0.05, 1.5, (n1, 1))
    s2 = rng.uniform(0.05, 1.5, (n2, 1))
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return dict(pc1=f(pc1), pc2=f(pc2), kp1=f(kp1), kp2=f(kp2), sigma1=f(s1), sigma2=f(s2), desc1=f(d1), desc2=f(d2), M=f(M))


def main():
    assert os.path.isdir(REF), "fixture generation needs /root/reference (build container only)"
    sys.path.insert(0, REF)
    from models.loss_utils import KeypointLoss, CorrespondenceLoss       # the reference's own classes
    from misc.poses import apply_transform
    gammas = dict(gamma_chamfer=1.0, gamma_p2p=1.0, gamma_c=1.0, gamma_k=1.0, beta=2.0, dist_th=0.5)   # make_losses defaults
    cases = {"a": (300, 280, 4000, 3800, 0.7, 0.05), "b": (64, 97, 700, 650, 0.3, 0.15), "c": (150, 150, 1500, 1500, 0.9, 0.02)}
    out = {}
    for name, (n1, n2, m1, m2, ov, nz) in cases.items():
        rng = np.random.default_rng(hash(name) % 1000 + 7 if False else {"a": 11, "b": 12, "c": 13}[name])
        p = make_pair(rng, n1, n2, m1, m2, ov, nz)
        t = {k: torch.from_numpy(v).clone() for k, v in p.items()}
        for k in ("kp1", "kp2", "sigma1", "sigma2", "desc1", "desc2"):
            t[k].requires_grad_(True)
        kl = KeypointLoss(gamma_chamfer=gammas["gamma_chamfer"], gamma_p2p=gammas["gamma_p2p"], prob_chamfer_loss=True,
                          p2p_loss=True, repeatability_dist_th=gammas["dist_th"])
        cl = CorrespondenceLoss(beta=gammas["beta"], dist_th=gammas["dist_th"])
        dist = torch.cdist(apply_transform(t["kp1"], t["M"]), t["kp2"])
        loss_k, met_k = kl(t["pc1"], t["kp1"], t["sigma1"], t["pc2"], t["kp2"], t["sigma2"], dist)
        loss_c, met_c = cl(t["desc1"], t["desc2"], dist)
        total = gammas["gamma_k"] * loss_k + gammas["gamma_c"] * loss_c
        total.backward()
        for k, v in p.items():
            out[f"{name}_{k}"] = v
        out[f"{name}_loss_keypoint"] = np.float32(loss_k.item())
        out[f"{name}_loss_correspondence"] = np.float32(loss_c.item())
        out[f"{name}_loss_total"] = np.float32(total.item())
        for k, v in {**met_k, **met_c}.items():
            out[f"{name}_metric_{k}"] = np.float64(v)
        for k in ("kp1", "kp2", "sigma1", "sigma2", "desc1", "desc2"):
            out[f"{name}_grad_{k}"] = t[k].grad.numpy().astype(np.float32)
        # the dense distance matrix the reference classes were fed (for the matrix-input entry points).  NOTE: torch.cdist's
        # default mode on these sizes is the matmul form (|a|^2+|b|^2-2ab), whose fp32 error on sub-metre distances is ~1e-3 m;
        # the matrix is stored so the matrix-input classes can be fed the very same numbers.
        out[f"{name}_dist"] = dist.detach().numpy().astype(np.float32)
        # run B: the same classes with that matrix as a leaf -> d loss / d dist, and the keypoint gradients of the chamfer part
        tb = {k: torch.from_numpy(v).clone() for k, v in p.items()}
        for k in ("kp1", "kp2", "sigma1", "sigma2", "desc1", "desc2"):
            tb[k].requires_grad_(True)
        dleaf = dist.detach().clone().requires_grad_(True)
        lk, _ = kl(tb["pc1"], tb["kp1"], tb["sigma1"], tb["pc2"], tb["kp2"], tb["sigma2"], dleaf)
        lc, _ = cl(tb["desc1"], tb["desc2"], dleaf)
        (gammas["gamma_k"] * lk + gammas["gamma_c"] * lc).backward()
        out[f"{name}_leaf_grad_dist"] = dleaf.grad.numpy().astype(np.float32)
        for k in ("kp1", "kp2", "sigma1", "sigma2", "desc1", "desc2"):
            out[f"{name}_leaf_grad_{k}"] = tb[k].grad.numpy().astype(np.float32)
        # run C: the reference classes in float64 (same fp32 inputs widened) = the algorithm without cdist's fp32 matmul noise;
        # the matrix-free driver (exact fp32 distances) is held to this one
        tc = {k: torch.from_numpy(v).double() for k, v in p.items()}
        for k in ("kp1", "kp2", "sigma1", "sigma2", "desc1", "desc2"):
            tc[k].requires_grad_(True)
        d64 = torch.cdist(apply_transform(tc["kp1"], tc["M"]), tc["kp2"])
        lk, mk = kl(tc["pc1"], tc["kp1"], tc["sigma1"], tc["pc2"], tc["kp2"], tc["sigma2"], d64)
        lc, mc = cl(tc["desc1"], tc["desc2"], d64)
        tot = gammas["gamma_k"] * lk + gammas["gamma_c"] * lc
        tot.backward()
        out[f"{name}_f64_loss_total"] = np.float64(tot.item())
        for k, v in {**mk, **mc}.items():
            out[f"{name}_f64_metric_{k}"] = np.float64(v)
        for k in ("kp1", "kp2", "sigma1", "sigma2", "desc1", "desc2"):
            out[f"{name}_f64_grad_{k}"] = tc[k].grad.numpy().astype(np.float32)
        # run C': float64 with the matrix as a leaf -> the keypoint gradients of the point-to-point term alone
        tl = {k: torch.from_numpy(v).double() for k, v in p.items()}
        for k in ("kp1", "kp2"):
            tl[k].requires_grad_(True)
        lk, _ = kl(tl["pc1"], tl["kp1"], tl["sigma1"], tl["pc2"], tl["kp2"], tl["sigma2"], d64.detach())
        (gammas["gamma_k"] * lk).backward()
        for k in ("kp1", "kp2"):
            out[f"{name}_f64_leaf_grad_{k}"] = tl[k].grad.numpy().astype(np.float32)
        print("   f64 total", tot.item(), "vs f32", total.item())
        print(name, "keypoint", loss_k.item(), "correspondence", loss_c.item(), {k: round(float(v), 4) for k, v in {**met_k, **met_c}.items()})
    out["gammas"] = np.array([gammas[k] for k in ("gamma_chamfer", "gamma_p2p", "gamma_c", "gamma_k", "beta", "dist_th")], np.float64)
    out["cases"] = np.array(sorted(cases))
    path = os.path.join(HERE, "local_losses.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
