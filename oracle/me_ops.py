"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path (egonn_amd/).

CPU restatement (numpy, fp32) of the MinkowskiEngine 0.5.4 operator subset that the
EgoNN descriptor-extraction path calls.  The reference
(/root/reference) contains no native code: every op below lives in the un-vendored
dependency MinkowskiEngine 0.5.4 (pinned only by prose in reference README.md:46,51),
which is absent from this image.  The semantics are therefore restated from the
published behaviour of ME 0.5.4 and anchored on the reference's own call sites:

  sparse_quantize      <- datasets/quantization.py:42,83
  batched_coordinates  <- eval/evaluate.py:333, datasets/dataset_utils.py:77
  stride-2 coord maps  <- models/minkgl.py:105 (MinkowskiConvolution k=2,s=2)
  conv k=5/3/2/1       <- models/minkgl.py:100,105,43,124 ; ME BasicBlock via layers/eca_block.py:58-63
  transposed conv      <- models/minkgl.py:39
  global avg pooling   <- layers/eca_block.py:16, layers/pooling.py:80
  broadcast mul        <- layers/eca_block.py:19,36

PARITY STATUS: **parity unpinned** for the sparse-conv primitive arithmetic — the
reference ships no tests, golden vectors or known-answer data, and
ME itself cannot be run here.  What IS pinned: the reference's own graph code
(models/*.py, layers/*.py, datasets/quantization.py) is executed on top of these
primitives (through oracle/MinkowskiEngine, this container only) to generate the
golden fixtures under tests/golden/, and the primitives are checked against
hand-derivable known-answer cases in tests/test_oracle.py.

Row order: ME's row order is hash-iteration order and not reproducible; everything
here is compared by coordinate, never by row.
"""
from __future__ import annotations

import numpy as np

_M = 1 << 16          # per-axis key radix
_O = 1 << 15          # per-axis offset so that negative coordinates encode


def encode_rows(c4: np.ndarray) -> np.ndarray:
    """(N,4) int [b,x,y,z] -> int64 key, monotone in (b,x,y,z) lexicographic order."""
    c = np.asarray(c4, dtype=np.int64)
    assert c.ndim == 2 and c.shape[1] == 4
    assert (np.abs(c[:, 1:]) < _O).all(), "coordinate out of oracle key range"
    return ((c[:, 0] * _M + (c[:, 1] + _O)) * _M + (c[:, 2] + _O)) * _M + (c[:, 3] + _O)


class _Index:
    """Exact coordinate -> row lookup (sorted keys + searchsorted)."""

    def __init__(self, c4: np.ndarray):
        k = encode_rows(c4)
        self.order = np.argsort(k, kind="stable")
        self.keys = k[self.order]
        if len(self.keys) > 1:
            assert (np.diff(self.keys) > 0).all(), "duplicate coordinates in a coordinate map"

    def lookup(self, c4: np.ndarray) -> np.ndarray:
        """row index of each query coordinate, or -1."""
        q = encode_rows(c4)
        if len(self.keys) == 0:
            return np.full(len(q), -1, dtype=np.int64)
        pos = np.searchsorted(self.keys, q)
        pos_c = np.minimum(pos, len(self.keys) - 1)
        hit = self.keys[pos_c] == q
        return np.where(hit, self.order[pos_c], -1)


# --------------------------------------------------------------------------------------
# A.1  utils.sparse_quantize
# --------------------------------------------------------------------------------------
def sparse_quantize(x: np.ndarray, quantization_size=None, return_index: bool = True):
    """floor(x / q) -> int32, unique voxels, index of the FIRST point of every voxel.

    Output order = increasing first-occurrence index.  The division
    is a true fp32 division (torch CPU `tensor / python_float`), pinned against torch in
    tests/test_oracle.py.
    """
    x = np.asarray(x, dtype=np.float32)
    if quantization_size is None or quantization_size == 1 or quantization_size == 1.0:
        d = np.floor(x)
    else:
        d = np.floor(x / np.float32(quantization_size))
    d = d.astype(np.int32)
    # unique rows, keep first occurrence
    _, first = np.unique(d, axis=0, return_index=True)
    first = np.sort(first)
    if return_index:
        return d[first], first.astype(np.int64)
    return d[first]


# --------------------------------------------------------------------------------------
# A.2  utils.batched_coordinates
# --------------------------------------------------------------------------------------
def batched_coordinates(coords_list) -> np.ndarray:
    out = []
    for b, c in enumerate(coords_list):
        c = np.asarray(c)
        bc = np.empty((c.shape[0], 4), dtype=np.int32)
        bc[:, 0] = b
        bc[:, 1:] = c
        out.append(bc)
    if not out:
        return np.zeros((0, 4), dtype=np.int32)
    return np.concatenate(out, axis=0)


# --------------------------------------------------------------------------------------
# A.4  strided coordinate map
# --------------------------------------------------------------------------------------
def stride_coords(c4: np.ndarray, new_stride: int) -> np.ndarray:
    """floor(c / ts) * ts per spatial axis (batch untouched), de-duplicated.

    Returned in first-appearance order (any order is legal; consumers join by coordinate).
    """
    c = np.asarray(c4, dtype=np.int64).copy()
    c[:, 1:] = np.floor_divide(c[:, 1:], new_stride) * new_stride
    _, first = np.unique(c, axis=0, return_index=True)
    first = np.sort(first)
    return c[first].astype(np.int32)


# --------------------------------------------------------------------------------------
# A.5  kernel offsets and kernel maps
# --------------------------------------------------------------------------------------
def kernel_offsets(k: int, tensor_stride: int) -> np.ndarray:
    """(k^3, 3) offsets.  Kernel index i decomposes with the FIRST spatial axis fastest
    (ix = i % k, iy = (i // k) % k, iz = i // k^2).  Odd k: centred; even k: 0..k-1.
    Offsets are multiples of the INPUT tensor stride."""
    idx = np.arange(k ** 3)
    ix, iy, iz = idx % k, (idx // k) % k, idx // (k * k)
    o = np.stack([ix, iy, iz], axis=1).astype(np.int64)
    if k % 2 == 1:
        o -= k // 2
    return o * tensor_stride


def kernel_map(in_c4: np.ndarray, out_c4: np.ndarray, k: int, in_stride: int):
    """For every kernel index i: (in_rows, out_rows) with C_in[in] == C_out[out] + off_i."""
    index = _Index(in_c4)
    out_c4 = np.asarray(out_c4, dtype=np.int64)
    maps = []
    for off in kernel_offsets(k, in_stride):
        q = out_c4.copy()
        q[:, 1:] += off
        # guard the oracle key range (queries one step outside are simply misses)
        ok = (np.abs(q[:, 1:]) < _O).all(axis=1)
        j = np.full(len(q), -1, dtype=np.int64)
        if ok.any():
            j[ok] = index.lookup(q[ok])
        o = np.nonzero(j >= 0)[0]
        maps.append((j[o], o))
    return maps


def conv_forward(feat_in: np.ndarray, kernel: np.ndarray, maps, n_out: int) -> np.ndarray:
    """out[o] = sum_i sum_{(j,o) in map_i} F[j] @ kernel[i]   (fp32, per-offset GEMM + scatter-add,
    the order ME itself uses)."""
    feat_in = np.asarray(feat_in, dtype=np.float32)
    kernel = np.asarray(kernel, dtype=np.float32)
    cout = kernel.shape[-1]
    out = np.zeros((n_out, cout), dtype=np.float32)
    for i, (j, o) in enumerate(maps):
        if len(j) == 0:
            continue
        # every output row appears at most once per offset -> plain fancy-index add is exact
        out[o] += feat_in[j] @ kernel[i]
    return out


def conv_transpose_forward(feat_in: np.ndarray, kernel: np.ndarray, maps, n_out: int) -> np.ndarray:
    """A.6: kernel map of the fine->coarse strided conv with in/out swapped.
    `maps` is the (fine_rows, coarse_rows) map of the forward strided conv;
    out[fine] = F[coarse] @ kernel[i]."""
    feat_in = np.asarray(feat_in, dtype=np.float32)
    kernel = np.asarray(kernel, dtype=np.float32)
    cout = kernel.shape[-1]
    out = np.zeros((n_out, cout), dtype=np.float32)
    for i, (fine, coarse) in enumerate(maps):
        if len(fine) == 0:
            continue
        out[fine] += feat_in[coarse] @ kernel[i]
    return out


# --------------------------------------------------------------------------------------
# A.8  global pooling / broadcast
# --------------------------------------------------------------------------------------
def batch_rows(c4: np.ndarray, batch_size: int | None = None):
    b = np.asarray(c4)[:, 0].astype(np.int64)
    if batch_size is None:
        batch_size = int(b.max()) + 1 if len(b) else 0
    return [np.nonzero(b == i)[0] for i in range(batch_size)]


def global_avg_pool(feat: np.ndarray, c4: np.ndarray, batch_size: int | None = None) -> np.ndarray:
    rows = batch_rows(c4, batch_size)
    feat = np.asarray(feat, dtype=np.float32)
    out = np.zeros((len(rows), feat.shape[1]), dtype=np.float32)
    for i, r in enumerate(rows):
        if len(r):
            out[i] = feat[r].sum(axis=0, dtype=np.float32) / np.float32(len(r))
    return out


def global_max_pool(feat: np.ndarray, c4: np.ndarray, batch_size: int | None = None) -> np.ndarray:
    rows = batch_rows(c4, batch_size)
    feat = np.asarray(feat, dtype=np.float32)
    out = np.zeros((len(rows), feat.shape[1]), dtype=np.float32)
    for i, r in enumerate(rows):
        if len(r):
            out[i] = feat[r].max(axis=0)
    return out


def broadcast_mul(feat: np.ndarray, c4: np.ndarray, g: np.ndarray) -> np.ndarray:
    b = np.asarray(c4)[:, 0].astype(np.int64)
    return np.asarray(feat, dtype=np.float32) * np.asarray(g, dtype=np.float32)[b]
