"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path (egonn_amd/).

Independent numpy/fp32 restatement of the reference's EgoNN descriptor-extraction graph on
top of oracle/me_ops.py.  Each function cites the reference lines it follows.  It is
validated in the build container against golden vectors produced by running the
reference's OWN graph code (tests/golden/make_golden.py) and then travels to the GPU box as
the checker for the HIP path (tests -m gpu, __graft_entry__.smoke(), bench.py cpu_baseline).

PARITY STATUS: graph wiring, channel plan, quantiser / keypoint formulas, GeM, ECA, heads
and the output dict are pinned against the reference's Python; the sparse-conv primitive
arithmetic underneath is "parity unpinned" (MinkowskiEngine is absent — see me_ops.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np

try:
    from . import me_ops as ops
except ImportError:  # oracle/ on sys.path
    import me_ops as ops  # type: ignore

F32 = np.float32
BN_EPS = F32(1e-5)

# reference models/model_factory.py:39-49
PLANES = [32, 64, 64, 128, 128, 128, 128]
GLOBAL_LEVELS, GLOBAL_CH, GLOBAL_DIM = [5, 6, 7], 128, 256
LOCAL_LEVELS, LOCAL_CH, LOCAL_DIM = [3, 4], 64, 128


# ----------------------------------------------------------------------------- quantisers
class CartesianQuantizer:
    """reference datasets/quantization.py:75-103"""

    def __init__(self, quant_step: float):
        self.quant_step = quant_step

    def __call__(self, pc: np.ndarray):
        assert pc.shape[1] == 3
        return ops.sparse_quantize(pc, quantization_size=self.quant_step, return_index=True)

    def dequantize(self, coords):
        return ((F32(0.5) + coords.astype(F32)) * F32(self.quant_step)).astype(F32)

    def keypoint_position(self, supervoxel_centers, stride, kp_offset):
        # quantization.py:93-103 : (C + 0.5) * q + offset * (stride * q) / 2
        c = (supervoxel_centers.astype(F32) + F32(0.5)) * F32(self.quant_step)
        size = np.asarray(stride, dtype=F32) * F32(self.quant_step)
        if kp_offset is None:
            return c.astype(F32)
        return (c + kp_offset.astype(F32) * size / F32(2.0)).astype(F32)


class PolarQuantizer:
    """reference datasets/quantization.py:22-72"""

    def __init__(self, quant_step):
        assert len(quant_step) == 3
        self.quant_step = np.asarray(quant_step, dtype=F32)
        self.theta_range = int(360.0 // float(self.quant_step[0]))

    def to_polar(self, pc: np.ndarray) -> np.ndarray:
        pc = np.asarray(pc, dtype=F32)
        # quantization.py:35 — `180. + atan2(y, x) * 180. / np.pi` evaluates left to right in fp32:
        # (atan2 * fp32(180)) / fp32(pi), then + 180
        theta = F32(180.0) + (np.arctan2(pc[:, 1], pc[:, 0]).astype(F32) * F32(180.0)) / F32(np.pi)
        dist = np.sqrt(pc[:, 0] ** 2 + pc[:, 1] ** 2).astype(F32)
        polar = np.stack([theta, dist, pc[:, 2]], axis=1).astype(F32)
        return (polar / self.quant_step).astype(F32)

    def __call__(self, pc: np.ndarray):
        assert pc.shape[1] == 3
        return ops.sparse_quantize(self.to_polar(pc), quantization_size=1.0, return_index=True)

    def to_cartesian(self, pc):
        theta = (F32(np.pi) * (pc[:, 0] - F32(180.0)) / F32(180.0)).astype(F32)
        x = np.cos(theta).astype(F32) * pc[:, 1]
        y = np.sin(theta).astype(F32) * pc[:, 1]
        return np.stack([x, y, pc[:, 2]], axis=1).astype(F32)

    def dequantize(self, coords):
        return self.to_cartesian(((F32(0.5) + coords.astype(F32)) * self.quant_step).astype(F32))

    def keypoint_position(self, supervoxel_centres, stride, kp_offset):
        # quantization.py:60-72
        c = (supervoxel_centres.astype(F32) + F32(0.5)) * self.quant_step
        size = np.asarray(stride, dtype=F32) * self.quant_step
        kp = (c + kp_offset.astype(F32) * size / F32(2.0)).astype(F32)
        return self.to_cartesian(kp)


# ----------------------------------------------------------------------------- row ops
def batchnorm_eval(x, sd, prefix):
    """MinkowskiBatchNorm = nn.BatchNorm1d(eps=1e-5) in eval mode (Appendix A.7)."""
    w, b = sd[prefix + ".bn.weight"], sd[prefix + ".bn.bias"]
    rm, rv = sd[prefix + ".bn.running_mean"], sd[prefix + ".bn.running_var"]
    inv = (F32(1.0) / np.sqrt(rv.astype(F32) + BN_EPS)).astype(F32)
    return ((x - rm) * inv * w + b).astype(F32)


def linear(x, sd, prefix):
    return (x @ sd[prefix + ".linear.weight"].T + sd[prefix + ".linear.bias"]).astype(F32)


def relu(x):
    return np.maximum(x, F32(0.0))


def softplus(x):
    # torch.nn.functional.softplus(beta=1, threshold=20)
    x = x.astype(F32)
    return np.where(x > F32(20.0), x, np.log1p(np.exp(np.minimum(x, F32(20.0))))).astype(F32)


def l2_normalize(x, eps=1e-12):
    n = np.sqrt((x * x).sum(axis=1, keepdims=True)).astype(F32)
    return (x / np.maximum(n, F32(eps))).astype(F32)


def conv1d_channels(y, w):
    """nn.Conv1d(1,1,k,padding=(k-1)//2,bias=False) along the channel axis of a (B,C) matrix
    (reference layers/eca_block.py:17,26) — cross-correlation with zero padding."""
    k = len(w)
    pad = (k - 1) // 2
    yp = np.pad(y, ((0, 0), (pad, pad)))
    out = np.zeros_like(y, dtype=F32)
    for j in range(k):
        out += F32(w[j]) * yp[:, j:j + y.shape[1]]
    return out.astype(F32)


def sigmoid(x):
    return (F32(1.0) / (F32(1.0) + np.exp(-x.astype(F32)))).astype(F32)


def morton3(c3: np.ndarray, bias: int = 1 << 15) -> np.ndarray:
    """Z-order key of (x,y,z) (x in the lowest bit) — tie-break order for keypoint selection."""
    v = (np.asarray(c3, dtype=np.int64) + bias).astype(np.uint64)
    key = np.zeros(len(v), dtype=np.uint64)
    for bit in range(16):
        for ax in range(3):
            key |= ((v[:, ax] >> np.uint64(bit)) & np.uint64(1)) << np.uint64(3 * bit + ax)
    return key


# ----------------------------------------------------------------------------- the graph
class SparseLevels:
    """Coordinate pyramid + cached kernel maps of one batch (what ME's CoordinateManager holds)."""

    def __init__(self, c4: np.ndarray, n_levels: int = 7):
        self.coords = {0: np.ascontiguousarray(c4, dtype=np.int32)}
        for l in range(1, n_levels + 1):
            self.coords[l] = ops.stride_coords(self.coords[l - 1], 1 << l)
        self.batch_size = int(c4[:, 0].max()) + 1 if len(c4) else 0
        self._maps = {}

    def kmap(self, lin: int, lout: int, k: int):
        key = (lin, lout, k)
        if key not in self._maps:
            self._maps[key] = ops.kernel_map(self.coords[lin], self.coords[lout], k, 1 << lin)
        return self._maps[key]

    def n(self, l):
        return len(self.coords[l])


class EgoNNOracle:
    """model_factory('egonn') + MinkGL.forward in eval mode (reference models/model_factory.py:31-76,
    models/minkgl.py:267-315)."""

    def __init__(self, state_dict: Dict[str, np.ndarray], quantizer):
        self.sd = {k: np.asarray(v) for k, v in state_dict.items()}
        self.quantizer = quantizer
        self.ignore_keypoint_regressor = False

    # -- reference layers/eca_block.py:56-73 + ME BasicBlock ctor (Appendix A.8)
    def _eca_block(self, x, lv: SparseLevels, level: int, prefix: str):
        sd = self.sd
        maps = lv.kmap(level, level, 3)
        n = lv.n(level)
        out = ops.conv_forward(x, sd[prefix + ".conv1.kernel"], maps, n)
        out = relu(batchnorm_eval(out, sd, prefix + ".norm1"))
        out = ops.conv_forward(out, sd[prefix + ".conv2.kernel"], maps, n)
        out = batchnorm_eval(out, sd, prefix + ".norm2")
        # ECALayer (eca_block.py:21-36)
        c4 = lv.coords[level]
        y = ops.global_avg_pool(out, c4, lv.batch_size)
        y = sigmoid(conv1d_channels(y, sd[prefix + ".eca.conv.weight"].reshape(-1)))
        out = ops.broadcast_mul(out, c4, y)
        if (prefix + ".downsample.0.kernel") in sd:
            res = (x @ sd[prefix + ".downsample.0.kernel"]).astype(F32)
            res = batchnorm_eval(res, sd, prefix + ".downsample.1")
        else:
            res = x
        return relu(out + res)

    # -- reference models/minkgl.py:136-153
    def trunk(self, feats, lv: SparseLevels):
        sd = self.sd
        x = ops.conv_forward(feats, sd["trunk.convs.0.kernel"], lv.kmap(0, 0, 5), lv.n(0))
        x = relu(batchnorm_eval(x, sd, "trunk.bn.0"))
        y = {}
        for i in range(1, len(PLANES) + 1):
            x = ops.conv_forward(x, sd[f"trunk.convs.{i}.kernel"], lv.kmap(i - 1, i, 2), lv.n(i))
            x = relu(batchnorm_eval(x, sd, f"trunk.bn.{i}"))
            x = self._eca_block(x, lv, i, f"trunk.blocks.{i}.0")
            y[i] = x
        return y

    # -- reference models/minkgl.py:46-60
    def head(self, x, lv: SparseLevels, name: str, levels: List[int]):
        sd = self.sd
        lo, hi = min(levels), max(levels)
        y = (x[hi] @ sd[f"{name}.conv1x1.{hi}.kernel"]).astype(F32)
        for level in range(hi - 1, lo - 1, -1):
            y = ops.conv_transpose_forward(y, sd[f"{name}.tconv.{level + 1}.kernel"],
                                           lv.kmap(level, level + 1, 2), lv.n(level))
            if level in levels:
                y = (y + x[level] @ sd[f"{name}.conv1x1.{level}.kernel"]).astype(F32)
        return y

    def _mlp(self, x, prefix):
        return linear(relu(linear(x, self.sd, prefix + ".net.0")), self.sd, prefix + ".net.2")

    # -- reference layers/pooling.py:82-86
    def gem(self, x, c4, batch_size):
        p = F32(self.sd["global_pooling.pooling.p"].reshape(-1)[0])
        t = np.power(np.maximum(x, F32(1e-6)), p).astype(F32)
        t = ops.global_avg_pool(t, c4, batch_size)
        return np.power(t, F32(1.0) / p).astype(F32)

    # -- reference layers/pooling.py:46-56 (MAC: per-sample max) and :59-69 (SPoC: per-sample mean)
    def mac(self, x, c4, batch_size):
        return ops.global_max_pool(np.asarray(x, dtype=F32), c4, batch_size).astype(F32)

    def spoc(self, x, c4, batch_size):
        return ops.global_avg_pool(np.asarray(x, dtype=F32), c4, batch_size).astype(F32)

    def forward(self, coords: np.ndarray, features: np.ndarray, disable_global_head=False,
                disable_local_head=False, return_internals=False, pool_method: str = "GeM"):
        c4 = np.asarray(coords, dtype=np.int32)
        lv = SparseLevels(c4)
        x = self.trunk(np.asarray(features, dtype=F32), lv)
        y = {}
        if not disable_global_head:
            g = self.head(x, lv, "global_head", GLOBAL_LEVELS)
            g = self._mlp(g, "global_descriptor_decoder")
            pool = {"GeM": self.gem, "MAC": self.mac, "SPoC": self.spoc}[pool_method]    # layers/pooling.py:13-43 PoolingWrapper
            y["global"] = pool(g, lv.coords[min(GLOBAL_LEVELS)], lv.batch_size)
        if not disable_local_head:
            lvl = min(LOCAL_LEVELS)
            xl = self.head(x, lv, "local_head", LOCAL_LEVELS)
            c_loc = lv.coords[lvl]
            rows = ops.batch_rows(c_loc, lv.batch_size)
            desc = l2_normalize(self._mlp(xl, "local_descriptor_decoder"))
            off = np.tanh(self._mlp(xl, "local_keypoint_regressor")).astype(F32)
            if self.ignore_keypoint_regressor:
                off = np.zeros_like(off)
            stride = [1 << lvl] * 3
            kp = self.quantizer.keypoint_position(c_loc[:, 1:], stride, off)
            sig = softplus(self._mlp(xl, "local_sigma_regressor"))
            y["descriptors"] = [desc[r] for r in rows]
            y["keypoints"] = [kp[r] for r in rows]
            y["sigma"] = [sig[r] for r in rows]
            y["keypoint_coords"] = [c_loc[r] for r in rows]   # join key (not in the reference dict)
        if return_internals:
            y["_levels"] = lv
            y["_trunk"] = x
        return y


# ----------------------------------------------------------------------------- evaluator slice
def select_keypoints(sigma: np.ndarray, kp_coords: np.ndarray, n_k: int = 128) -> np.ndarray:
    """reference eval/evaluate.py:352-361: the n_k keypoints with the lowest sigma in increasing
    order.  torch.topk's tie order is unspecified; this build's rule is ties
    broken by the Z-order key of the super-voxel coordinate."""
    s = np.asarray(sigma, dtype=F32).reshape(-1)
    n_k = min(len(s), n_k)
    tie = morton3(kp_coords[:, 1:])
    order = np.lexsort((tie, s))
    return order[:n_k]


def compute_embedding(oracle: EgoNNOracle, pc: np.ndarray, n_k: int = 128):
    """reference eval/evaluate.py:327-350 for one scan."""
    coords, _ = oracle.quantizer(pc)
    bc = ops.batched_coordinates([coords])
    y = oracle.forward(bc, np.ones((len(bc), 1), dtype=F32))
    idx = select_keypoints(y["sigma"][0], y["keypoint_coords"][0], n_k)
    return y["global"], y["keypoints"][0][idx], y["descriptors"][0][idx], y["keypoint_coords"][0][idx]


def compute_embedding_with_sigma(oracle: EgoNNOracle, pc: np.ndarray, n_k: int = 128):
    """compute_embedding plus the (ascending) sigmas of the selected keypoints: tests use the gaps between them to tell a
    legitimate swap of two near-equal saliencies from a wrong selection."""
    coords, _ = oracle.quantizer(pc)
    bc = ops.batched_coordinates([coords])
    y = oracle.forward(bc, np.ones((len(bc), 1), dtype=F32))
    idx = select_keypoints(y["sigma"][0], y["keypoint_coords"][0], n_k)
    return (y["global"], y["keypoints"][0][idx], y["descriptors"][0][idx], y["keypoint_coords"][0][idx],
            y["sigma"][0][idx].reshape(-1))


# ----------------------------------------------------------------------------- MinkLoc / MinkLoc3D (MinkFPN + GeM)
class MinkLocOracle:
    """reference models/minkfpn.py:65-93 (MinkFPN.forward) + GeM, as used by models/minkloc.py:44-61 and
    third_party/minkloc3d/minkloc.py:22-33.  `gem_key` is 'pooling.p' (MinkLoc3D) or 'pooling.pooling.p' (MinkLoc);
    blocks are ME BasicBlocks, with the ECA gate when the state_dict holds '...eca.conv.weight'."""

    def __init__(self, state_dict: Dict[str, np.ndarray], planes=(32, 64, 64), layers=(1, 1, 1), num_top_down=1):
        self.sd = {k: np.asarray(v) for k, v in state_dict.items()}
        self.planes, self.layers, self.num_top_down = list(planes), list(layers), num_top_down
        self.gem_key = "pooling.p" if "pooling.p" in self.sd else "pooling.pooling.p"

    def _block(self, x, lv: SparseLevels, level: int, prefix: str):
        sd = self.sd
        maps = lv.kmap(level, level, 3)
        n = lv.n(level)
        out = relu(batchnorm_eval(ops.conv_forward(x, sd[prefix + ".conv1.kernel"], maps, n), sd, prefix + ".norm1"))
        out = batchnorm_eval(ops.conv_forward(out, sd[prefix + ".conv2.kernel"], maps, n), sd, prefix + ".norm2")
        if (prefix + ".eca.conv.weight") in sd:
            c4 = lv.coords[level]
            y = ops.global_avg_pool(out, c4, lv.batch_size)
            y = sigmoid(conv1d_channels(y, sd[prefix + ".eca.conv.weight"].reshape(-1)))
            out = ops.broadcast_mul(out, c4, y)
        if (prefix + ".downsample.0.kernel") in sd:
            res = batchnorm_eval((x @ sd[prefix + ".downsample.0.kernel"]).astype(F32), sd, prefix + ".downsample.1")
        else:
            res = x
        return relu(out + res)

    def backbone(self, coords, features):
        sd = self.sd
        nb = len(self.planes)
        lv = SparseLevels(np.asarray(coords, dtype=np.int32), n_levels=nb)
        x = ops.conv_forward(np.asarray(features, dtype=F32), sd["backbone.conv0.kernel"], lv.kmap(0, 0, 5), lv.n(0))
        x = relu(batchnorm_eval(x, sd, "backbone.bn0"))
        fmaps = []
        if self.num_top_down == nb:
            fmaps.append(x)
        for ndx in range(nb):
            x = ops.conv_forward(x, sd[f"backbone.convs.{ndx}.kernel"], lv.kmap(ndx, ndx + 1, 2), lv.n(ndx + 1))
            x = relu(batchnorm_eval(x, sd, f"backbone.bn.{ndx}"))
            for b in range(self.layers[ndx]):
                x = self._block(x, lv, ndx + 1, f"backbone.blocks.{ndx}.{b}")
            if nb - 1 - self.num_top_down <= ndx < nb - 1:
                fmaps.append(x)
        x = (x @ sd["backbone.conv1x1.0.kernel"]).astype(F32)
        level = nb
        for ndx in range(self.num_top_down):
            x = ops.conv_transpose_forward(x, sd[f"backbone.tconvs.{ndx}.kernel"], lv.kmap(level - 1, level, 2),
                                           lv.n(level - 1))
            level -= 1
            x = (x + fmaps[-ndx - 1] @ sd[f"backbone.conv1x1.{ndx + 1}.kernel"]).astype(F32)
        return lv, level, x

    def forward(self, coords, features):
        lv, level, x = self.backbone(coords, features)
        p = F32(self.sd[self.gem_key].reshape(-1)[0])
        t = np.power(np.maximum(x, F32(1e-6)), p).astype(F32)
        t = ops.global_avg_pool(t, lv.coords[level], lv.batch_size)
        return {"global": np.power(t, F32(1.0) / p).astype(F32), "_coords": lv.coords[level], "_feats": x}


# ----------------------------------------------------------------------------- batch-hard triplet loss
def batch_hard_triplet_loss(emb: np.ndarray, pos_mask: np.ndarray, neg_mask: np.ndarray, margin: float):
    """reference models/loss.py:114-172 — miner (in-tree: get_max_per_row / get_min_per_row :132-143) + the
    pytorch_metric_learning pieces it calls, restated per SURVEY.md Appendix A.9 (parity unpinned for those):
    LpDistance(p=2), TripletMarginLoss(margin, swap=True), AvgNonZeroReducer.  Returns (loss, stats, (a, p, n))."""
    e = np.asarray(emb, dtype=np.float64)
    n = len(e)
    diff = e[:, None, :] - e[None, :, :]
    D = np.sqrt((diff * diff).sum(-1))
    pm, nm = np.asarray(pos_mask, bool), np.asarray(neg_mask, bool)
    mp = np.where(pm, D, 0.0)
    mn = np.where(nm, D, np.inf)
    hp_idx, hn_idx = mp.argmax(1), mn.argmin(1)
    hp, hn = mp.max(1), mn.min(1)
    keep = pm.any(1) & nm.any(1)
    a = np.arange(n)[keep]
    p, q = hp_idx[keep], hn_idx[keep]
    d_ap, d_an = D[a, p], np.minimum(D[a, q], D[p, q])
    li = np.maximum(d_ap - d_an + margin, 0.0)
    nz = int((li > 0).sum())
    loss = float(li[li > 0].mean()) if nz else 0.0
    stats = {"loss": loss, "num_triplets": int(len(a)), "num_non_zero_triplets": nz,
             "avg_embedding_norm": float(np.linalg.norm(e, axis=1).mean()),
             "mean_pos_pair_dist": float(hp.mean()), "max_pos_pair_dist": float(hp.max()),
             "min_pos_pair_dist": float(hp.min()), "mean_neg_pair_dist": float(hn.mean()),
             "max_neg_pair_dist": float(hn.max()), "min_neg_pair_dist": float(hn.min())}
    return loss, stats, (a, p, q)
