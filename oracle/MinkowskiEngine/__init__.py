"""ORACLE — TEST INFRASTRUCTURE ONLY (fixture generation in the build container).

A module *named* MinkowskiEngine exposing exactly the symbols the reference graph touches
(SURVEY.md Appendix D), implemented on oracle/me_ops.py (numpy index logic) + torch CPU
ops for the arithmetic (so autograd works).  It exists so that the reference's OWN Python
(models/*.py, layers/*.py, datasets/quantization.py) can be executed here to generate the
golden vectors in tests/golden/ — see tests/golden/make_golden.py.

This is NOT MinkowskiEngine and makes no claim to be: ME 0.5.4 is absent from the image,
so the primitive semantics are "parity unpinned" (see oracle/me_ops.py header).  Nothing
in egonn_amd/ imports this package.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

try:                                   # imported as oracle.MinkowskiEngine
    from .. import me_ops as _ops      # type: ignore
except ImportError:                    # imported as top-level MinkowskiEngine (oracle/ on sys.path)
    import me_ops as _ops              # type: ignore

__version__ = "0.5.4-oracle-standin"


# ---------------------------------------------------------------------------------------
# coordinate manager
# ---------------------------------------------------------------------------------------
class CoordinateMapKey:
    def __init__(self, stride: int):
        self.stride = int(stride)

    def get_tensor_stride(self):
        return [self.stride] * 3

    def __eq__(self, other):
        return isinstance(other, CoordinateMapKey) and other.stride == self.stride

    def __hash__(self):
        return hash(self.stride)

    def __repr__(self):
        return f"CoordinateMapKey(stride={self.stride})"


_ORIGIN = -1  # pseudo-stride of the "reduced to origin" map used by global pooling


class CoordinateManager:
    def __init__(self):
        self.coords = {}      # stride -> (N,4) int32 numpy
        self.kmaps = {}       # (in_stride, out_stride, k) -> list[(in_rows, out_rows)]
        self.batch_size = 0

    def insert(self, stride: int, c4: np.ndarray):
        self.coords[stride] = np.ascontiguousarray(c4, dtype=np.int32)
        if stride == 1:
            self.batch_size = int(c4[:, 0].max()) + 1 if len(c4) else 0

    def get_coords(self, stride: int) -> np.ndarray:
        return self.coords[stride]

    def stride_map(self, in_stride: int, out_stride: int) -> np.ndarray:
        if out_stride not in self.coords:
            self.coords[out_stride] = _ops.stride_coords(self.coords[in_stride], out_stride)
        return self.coords[out_stride]

    def kernel_map(self, in_stride: int, out_stride: int, k: int):
        key = (in_stride, out_stride, k)
        if key not in self.kmaps:
            self.kmaps[key] = _ops.kernel_map(self.coords[in_stride], self.coords[out_stride], k, in_stride)
        return self.kmaps[key]


class SparseTensor:
    def __init__(self, features, coordinates=None, coordinate_manager=None, coordinate_map_key=None,
                 tensor_stride=1, **kwargs):
        assert isinstance(features, torch.Tensor) and features.dim() == 2
        if coordinate_manager is None:
            assert coordinates is not None
            c = coordinates.detach().cpu().numpy() if isinstance(coordinates, torch.Tensor) else np.asarray(coordinates)
            assert c.ndim == 2 and c.shape[1] == 4
            # RANDOM_SUBSAMPLE de-duplication: identity for unique input (Appendix A.3)
            keys = _ops.encode_rows(c)
            _, first = np.unique(keys, return_index=True)
            if len(first) != len(keys):
                first = np.sort(first)
                c = c[first]
                features = features[torch.from_numpy(first)]
            coordinate_manager = CoordinateManager()
            coordinate_manager.insert(1, c)
            coordinate_map_key = CoordinateMapKey(1)
        assert coordinate_map_key is not None
        self._F = features
        self.coordinate_manager = coordinate_manager
        self.coordinate_map_key = coordinate_map_key

    # -- attributes used by the reference --
    @property
    def F(self):
        return self._F

    @property
    def C(self):
        return torch.from_numpy(self.coordinate_manager.get_coords(self.coordinate_map_key.stride))

    @property
    def shape(self):
        return self._F.shape

    @property
    def tensor_stride(self):
        return self.coordinate_map_key.get_tensor_stride()

    @property
    def device(self):
        return self._F.device

    @property
    def _batchwise_row_indices(self):
        c = self.coordinate_manager.get_coords(self.coordinate_map_key.stride)
        return [torch.from_numpy(r) for r in _ops.batch_rows(c, self.coordinate_manager.batch_size)]

    @property
    def decomposed_features(self):
        return [self._F[r] for r in self._batchwise_row_indices]

    def _same_map(self, other):
        assert isinstance(other, SparseTensor)
        assert other.coordinate_manager is self.coordinate_manager
        assert other.coordinate_map_key == self.coordinate_map_key, "stand-in only supports same-key add"

    def __add__(self, other):
        self._same_map(other)
        return SparseTensor(self._F + other._F, coordinate_manager=self.coordinate_manager,
                            coordinate_map_key=self.coordinate_map_key)

    def __iadd__(self, other):
        self._same_map(other)
        self._F = self._F + other._F
        return self

    def dim(self):
        return self._F.dim()


def _like(x: SparseTensor, feats: torch.Tensor, stride=None) -> SparseTensor:
    key = x.coordinate_map_key if stride is None else CoordinateMapKey(stride)
    return SparseTensor(feats, coordinate_manager=x.coordinate_manager, coordinate_map_key=key)


# ---------------------------------------------------------------------------------------
# convolutions
# ---------------------------------------------------------------------------------------
class MinkowskiConvolutionBase(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, is_transpose=False, expand_coordinates=False, dimension=-1):
        super().__init__()
        assert dimension == 3 and dilation == 1
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.is_transpose = int(kernel_size), int(stride), is_transpose
        kv = self.kernel_size ** 3
        self.kernel_volume = kv
        shape = (in_channels, out_channels) if kv == 1 else (kv, in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(*shape))
        self.bias = nn.Parameter(torch.empty(1, out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        with torch.no_grad():
            n = (self.out_channels if self.is_transpose else self.in_channels) * self.kernel_volume
            stdv = 1.0 / np.sqrt(n)
            self.kernel.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.uniform_(-stdv, stdv)

    def _apply_maps(self, x, maps, n_out, swap):
        out = torch.zeros((n_out, self.out_channels), dtype=x.F.dtype)
        for i, (a, b) in enumerate(maps):
            if len(a) == 0:
                continue
            src, dst = (b, a) if swap else (a, b)
            out = out.index_add(0, torch.from_numpy(dst), x.F[torch.from_numpy(src)] @ self.kernel[i])
        if self.bias is not None:
            out = out + self.bias
        return out


class MinkowskiConvolution(MinkowskiConvolutionBase):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, expand_coordinates=False, dimension=None):
        super().__init__(in_channels, out_channels, kernel_size, stride, dilation, bias,
                         is_transpose=False, dimension=dimension)

    def forward(self, x: SparseTensor) -> SparseTensor:
        cm, t = x.coordinate_manager, x.coordinate_map_key.stride
        if self.kernel_volume == 1:
            assert self.stride == 1
            out = x.F @ self.kernel
            if self.bias is not None:
                out = out + self.bias
            return _like(x, out)
        out_stride = t * self.stride
        out_c = cm.stride_map(t, out_stride) if self.stride > 1 else cm.get_coords(t)
        maps = cm.kernel_map(t, out_stride, self.kernel_size)
        return _like(x, self._apply_maps(x, maps, len(out_c), swap=False), stride=out_stride)


class MinkowskiConvolutionTranspose(MinkowskiConvolutionBase):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False,
                 kernel_generator=None, expand_coordinates=False, dimension=None):
        super().__init__(in_channels, out_channels, kernel_size, stride, dilation, bias,
                         is_transpose=True, dimension=dimension)

    def forward(self, x: SparseTensor) -> SparseTensor:
        cm, t = x.coordinate_manager, x.coordinate_map_key.stride
        assert self.stride > 1 and t % self.stride == 0
        out_stride = t // self.stride
        # Appendix A.6: the finer coordinate map already exists (made on the way down) and is reused
        assert out_stride in cm.coords, "stand-in supports transposed conv onto a cached coordinate map only"
        fine_c = cm.get_coords(out_stride)
        maps = cm.kernel_map(out_stride, t, self.kernel_size)   # fine -> coarse map of the strided conv
        return _like(x, self._apply_maps(x, maps, len(fine_c), swap=True), stride=out_stride)


# ---------------------------------------------------------------------------------------
# row-wise wrappers
# ---------------------------------------------------------------------------------------
class MinkowskiBatchNorm(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum, affine=affine,
                                 track_running_stats=track_running_stats)

    def forward(self, x):
        return _like(x, self.bn(x.F))


class MinkowskiLinear(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features, bias=bias)

    def forward(self, x):
        return _like(x, self.linear(x.F))


class _Elementwise(nn.Module):
    fn = None

    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, x):
        return _like(x, type(self).fn(x.F))


class MinkowskiReLU(_Elementwise):
    fn = staticmethod(torch.relu)


class MinkowskiSigmoid(_Elementwise):
    fn = staticmethod(torch.sigmoid)


class MinkowskiTanh(_Elementwise):
    fn = staticmethod(torch.tanh)


class MinkowskiSoftplus(_Elementwise):
    fn = staticmethod(nn.functional.softplus)


# ---------------------------------------------------------------------------------------
# pooling / broadcast
# ---------------------------------------------------------------------------------------
def _pool(x: SparseTensor, mode: str) -> SparseTensor:
    cm = x.coordinate_manager
    rows = x._batchwise_row_indices
    outs = []
    for r in rows:
        f = x.F[r]
        if mode == "avg":
            outs.append(f.sum(dim=0) / max(len(r), 1))
        else:
            outs.append(f.max(dim=0).values)
    out = torch.stack(outs, dim=0)
    if _ORIGIN not in cm.coords:
        oc = np.zeros((len(rows), 4), dtype=np.int32)
        oc[:, 0] = np.arange(len(rows))
        cm.coords[_ORIGIN] = oc
    return _like(x, out, stride=_ORIGIN)


class MinkowskiGlobalPooling(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, x):
        return _pool(x, "avg")


class MinkowskiGlobalAvgPooling(MinkowskiGlobalPooling):
    pass


class MinkowskiGlobalSumPooling(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()


class MinkowskiGlobalMaxPooling(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, x):
        return _pool(x, "max")


class MinkowskiAvgPooling(nn.Module):
    """ctor only (models/resnet.py:53); never called on the EgoNN / MinkFPN path."""

    def __init__(self, *args, **kwargs):
        super().__init__()


class MinkowskiBroadcastMultiplication(nn.Module):
    def forward(self, x: SparseTensor, g: SparseTensor) -> SparseTensor:
        c = x.coordinate_manager.get_coords(x.coordinate_map_key.stride)
        b = torch.from_numpy(c[:, 0].astype(np.int64))
        return _like(x, x.F * g.F[b])


class _Functional:
    @staticmethod
    def normalize(x: SparseTensor, p=2, dim=1, eps=1e-12):
        return _like(x, nn.functional.normalize(x.F, p=p, dim=dim, eps=eps))

    @staticmethod
    def relu(x):
        return _like(x, torch.relu(x.F))


MinkowskiFunctional = _Functional


# ---------------------------------------------------------------------------------------
# utils
# ---------------------------------------------------------------------------------------
class _Utils:
    @staticmethod
    def sparse_quantize(coordinates, features=None, labels=None, ignore_label=-100, return_index=False,
                        return_inverse=False, return_maps_only=False, quantization_size=None, device="cpu"):
        assert features is None and labels is None and not return_inverse and not return_maps_only
        is_t = isinstance(coordinates, torch.Tensor)
        x = coordinates.detach().cpu().numpy() if is_t else np.asarray(coordinates)
        d, idx = _ops.sparse_quantize(x, quantization_size, return_index=True)
        if is_t:
            d, idx = torch.from_numpy(d), torch.from_numpy(idx)
        return (d, idx) if return_index else d

    @staticmethod
    def batched_coordinates(coords, dtype=torch.int32, device=None):
        arr = [c.detach().cpu().numpy() if isinstance(c, torch.Tensor) else np.asarray(c) for c in coords]
        return torch.from_numpy(_ops.batched_coordinates(arr)).to(dtype)

    @staticmethod
    def kaiming_normal_(tensor, a=0, mode="fan_in", nonlinearity="leaky_relu"):
        # fan_out = Cout * K for (K, Cin, Cout) kernels (Appendix A.8)
        if tensor.dim() == 3:
            kv, cin, cout = tensor.shape
        else:
            kv, (cin, cout) = 1, tensor.shape
        fan = cin * kv if mode == "fan_in" else cout * kv
        gain = nn.init.calculate_gain(nonlinearity, a)
        std = gain / np.sqrt(fan)
        with torch.no_grad():
            return tensor.normal_(0, std)


utils = _Utils
