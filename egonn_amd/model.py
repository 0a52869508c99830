"""EgoNN model with the reference's Python surface, executed by libegonn_hip.

`model_factory(model_params)` returns a `MinkGL` module whose
  * `state_dict()` has exactly the reference's keys/shapes (SURVEY.md Appendix B;
    tests/golden/egonn_state_dict_shapes.json was extracted from the reference model),
  * `forward(batch)` takes {'coords': (N,4) int32 [b,x,y,z], 'features': (N,1) f32} and returns
    {'global': (B,256), 'descriptors': [..]*B, 'keypoints': [..]*B, 'sigma': [..]*B}
    (reference models/minkgl.py:267-315),
  * kwargs `disable_global_head` / `disable_local_head`, attribute `ignore_keypoint_regressor`,
    `print_info()` behave as in the reference.

The nn.Module tree only HOLDS parameters (so that load_state_dict / .to() / optimisers work); all
arithmetic runs in the HIP library.  There is no PyTorch fallback.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .quantization import Quantizer

PLANES = [32, 64, 64, 128, 128, 128, 128]          # reference models/model_factory.py:40
GLOBAL_LEVELS, GLOBAL_CH, GLOBAL_DIM = [5, 6, 7], 128, 256
LOCAL_LEVELS, LOCAL_CH, LOCAL_DIM = [3, 4], 64, 128


# ----------------------------------------------------------------------------- parameter holders
class SparseConv(nn.Module):
    """holds `.kernel` like ME.MinkowskiConvolution(Transpose): (K,Cin,Cout) or (Cin,Cout) for 1x1."""

    def __init__(self, cin, cout, kernel_size, transpose=False):
        super().__init__()
        kv = kernel_size ** 3
        self.kernel_size, self.transpose = kernel_size, transpose
        shape = (cin, cout) if kv == 1 else (kv, cin, cout)
        self.kernel = nn.Parameter(torch.empty(*shape))
        with torch.no_grad():                       # ME default init: U(-1/sqrt(n), 1/sqrt(n))
            n = (cout if transpose else cin) * kv
            self.kernel.uniform_(-1.0 / np.sqrt(n), 1.0 / np.sqrt(n))


class BatchNorm(nn.Module):
    """holds `.bn` like ME.MinkowskiBatchNorm."""

    def __init__(self, c):
        super().__init__()
        self.bn = nn.BatchNorm1d(c, eps=1e-5, momentum=0.1)


class Linear(nn.Module):
    """holds `.linear` like ME.MinkowskiLinear."""

    def __init__(self, cin, cout):
        super().__init__()
        self.linear = nn.Linear(cin, cout)


class _NoParams(nn.Module):
    """placeholder keeping nn.Sequential indices aligned with the reference (ReLU/Tanh/Softplus slots)."""


class ECALayer(nn.Module):
    """reference layers/eca_block.py:11-20"""

    def __init__(self, channels, gamma=2, b=1):
        super().__init__()
        t = int(abs((np.log2(channels) + b) / gamma))
        k_size = t if t % 2 else t + 1
        self.conv = nn.Conv1d(1, 1, kernel_size=k_size, padding=(k_size - 1) // 2, bias=False)


class ECABasicBlock(nn.Module):
    """reference layers/eca_block.py:39-54 + ME BasicBlock ctor"""
    expansion = 1

    def __init__(self, inplanes, planes, downsample=None):
        super().__init__()
        self.conv1 = SparseConv(inplanes, planes, 3)
        self.norm1 = BatchNorm(planes)
        self.conv2 = SparseConv(planes, planes, 3)
        self.norm2 = BatchNorm(planes)
        self.downsample = downsample
        self.eca = ECALayer(planes, gamma=2, b=1)


class MinkTrunk(nn.Module):
    """reference models/minkgl.py:68-134"""

    def __init__(self, in_channels: int, planes: List[int], conv0_kernel_size: int = 5):
        super().__init__()
        self.planes = planes
        self.convs, self.bn, self.blocks = nn.ModuleDict(), nn.ModuleDict(), nn.ModuleDict()
        inplanes = planes[0]
        self.convs['0'] = SparseConv(in_channels, inplanes, conv0_kernel_size)
        self.bn['0'] = BatchNorm(inplanes)
        for ndx, plane in enumerate(planes):
            self.convs[str(ndx + 1)] = SparseConv(inplanes, inplanes, 2)
            self.bn[str(ndx + 1)] = BatchNorm(inplanes)
            downsample = None
            if inplanes != plane:
                downsample = nn.Sequential(SparseConv(inplanes, plane, 1), BatchNorm(plane))
            self.blocks[str(ndx + 1)] = nn.Sequential(ECABasicBlock(inplanes, plane, downsample))
            inplanes = plane
        self.weight_initialization()

    def weight_initialization(self):
        # reference models/minkgl.py:112-118 (kaiming_normal_, mode='fan_out' = Cout*K, relu gain)
        for m in self.modules():
            if isinstance(m, SparseConv) and not m.transpose:
                k = m.kernel
                fan_out = k.shape[-1] * (k.shape[0] if k.dim() == 3 else 1)
                with torch.no_grad():
                    k.normal_(0, np.sqrt(2.0 / fan_out))
            if isinstance(m, BatchNorm):
                nn.init.constant_(m.bn.weight, 1)
                nn.init.constant_(m.bn.bias, 0)


class MinkHead(nn.Module):
    """reference models/minkgl.py:14-44"""

    def __init__(self, in_levels: List[int], in_channels: List[int], out_channels: int):
        super().__init__()
        assert len(in_levels) > 0 and len(in_levels) == len(in_channels)
        self.in_levels, self.in_channels, self.out_channels = in_levels, in_channels, out_channels
        self.min_level, self.max_level = min(in_levels), max(in_levels)
        assert self.min_level > 0
        self.conv1x1, self.tconv = nn.ModuleDict(), nn.ModuleDict()
        for lvl in range(self.min_level + 1, self.max_level + 1):
            self.tconv[str(lvl)] = SparseConv(out_channels, out_channels, 2, transpose=True)
        for lvl, ch in zip(in_levels, in_channels):
            self.conv1x1[str(lvl)] = SparseConv(ch, out_channels, 1)


class _MLP(nn.Module):
    def __init__(self, cin, mid, cout, tail: bool):
        super().__init__()
        mods = [Linear(cin, mid), _NoParams(), Linear(mid, cout)]
        if tail:
            mods.append(_NoParams())
        self.net = nn.Sequential(*mods)


class KeypointRegressor(_MLP):          # reference models/minkgl.py:175-185
    def __init__(self, in_channels, reduction=2):
        super().__init__(in_channels, in_channels // reduction, 3, tail=True)


class SigmaRegressor(_MLP):             # reference models/minkgl.py:188-204
    def __init__(self, in_channels, reduction=2):
        super().__init__(in_channels, in_channels // reduction, 1, tail=True)


class DescriptorDecoder(_MLP):          # reference models/minkgl.py:207-225
    def __init__(self, in_channels, out_channels, normalize=True):
        super().__init__(in_channels, out_channels + (in_channels - out_channels) // 2, out_channels, tail=False)
        self.normalize = normalize


class GeM(nn.Module):                   # reference layers/pooling.py:72-86
    def __init__(self, input_dim, p=3, eps=1e-6):
        super().__init__()
        self.input_dim = self.output_dim = input_dim
        self.p = nn.Parameter(torch.ones(1) * p)
        self.eps = eps


class MAC(nn.Module):                   # reference layers/pooling.py:46-56 (MinkowskiGlobalMaxPooling; no parameters)
    def __init__(self, input_dim):
        super().__init__()
        self.input_dim = self.output_dim = input_dim


class SPoC(nn.Module):                  # reference layers/pooling.py:59-69 (MinkowskiGlobalAvgPooling; no parameters)
    def __init__(self, input_dim):
        super().__init__()
        self.input_dim = self.output_dim = input_dim


class PoolingWrapper(nn.Module):        # reference layers/pooling.py:13-43
    def __init__(self, pool_method, in_dim, output_dim):
        super().__init__()
        if pool_method not in ('GeM', 'MAC', 'SPoC'):
            raise NotImplementedError(f'pooling method {pool_method!r}: the MI355X path implements GeM (the egonn '
                                      f'configuration, models/model_factory.py:73-76), MAC and SPoC; NetVLAD is unused '
                                      f'by the egonn configuration')
        assert in_dim == output_dim
        self.pool_method, self.in_dim, self.output_dim = pool_method, in_dim, output_dim
        self.pooling = {'GeM': GeM, 'MAC': MAC, 'SPoC': SPoC}[pool_method](input_dim=in_dim)


# ----------------------------------------------------------------------------- the model
class MinkGL(nn.Module):
    """reference models/minkgl.py:228-334"""

    def __init__(self, trunk: MinkTrunk, local_head: MinkHead = None, local_descriptor_size: int = None,
                 local_normalize: bool = True, global_head: MinkHead = None, global_descriptor_size: int = None,
                 global_pool_method: str = 'GeM', global_normalize: bool = False, quantizer: Quantizer = None):
        assert quantizer is not None
        super().__init__()
        self.trunk = trunk
        self.global_head = global_head
        self.global_pool_method = global_pool_method
        self.global_channels = global_head.out_channels
        self.global_pooling = PoolingWrapper(global_pool_method, self.global_channels, self.global_channels)
        self.global_normalize = global_normalize
        self.global_descriptor_size = global_descriptor_size
        self.global_descriptor_decoder = DescriptorDecoder(self.global_channels, global_descriptor_size,
                                                           normalize=False)
        self.local_head = local_head
        if local_head is not None:
            self.local_descriptor_size = local_descriptor_size
            self.local_normalize = local_normalize
            n = local_head.out_channels
            self.local_keypoint_regressor = KeypointRegressor(n, reduction=2)
            self.local_sigma_regressor = SigmaRegressor(n, reduction=2)
            self.local_descriptor_decoder = DescriptorDecoder(n, local_descriptor_size, normalize=local_normalize)
        self.quantizer = quantizer
        self.ignore_keypoint_regressor = False
        if global_normalize or (local_head is not None and not local_normalize) or local_head is None:
            raise NotImplementedError("the MI355X path implements the 'egonn' configuration "
                                      "(global_normalize=False, local_normalize=True, with local head)")
        # --- HIP side
        self._handle = None
        self._ctx = None
        self._registered = None
        self.coord_bits = 16

    # ------------------------------------------------------------------ HIP plumbing
    def _device(self) -> torch.device:
        dev = next(self.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError("egonn_amd.MinkGL runs on MI355X only: move the model to a HIP device "
                               "(`model.to('cuda')`); there is no CPU fallback.")
        return dev

    def context(self, slot: int = 0) -> _lib.Context:
        """egonn_ctx number `slot` of this model's device (one per batch in flight: each owns its plan+workspace)."""
        dev = self._device()
        if self._ctx is None or not isinstance(self._ctx, dict):
            self._ctx = {}
        c = self._ctx.get(slot)
        if c is None or c.device != dev:
            c = _lib.Context(dev, coord_bits=self.coord_bits)
            self._ctx[slot] = c
        return c

    def _float_state(self):
        for k, v in self.state_dict(keep_vars=True).items():
            if v.dtype == torch.float32:
                yield k, v

    def _sync_weights(self):
        """(Re)register weights with the HIP model when any tensor moved or was written to."""
        # (BatchNorm running statistics are updated by the HIP kernels through raw pointers, which does not bump their
        #  tensor version; every such update increments num_batches_tracked with a torch op, so its version is part of the
        #  signature and a train-mode forward always invalidates the folded scale/shift)
        sig = tuple((k, v.data_ptr(), v._version) for k, v in self.state_dict(keep_vars=True).items())
        if self._handle is not None and sig == self._registered:
            return
        if self._handle is None:
            self._handle = _lib.ModelHandle()
        for k, v in self._float_state():
            t = v.detach()
            if not t.is_contiguous():
                raise RuntimeError(f"parameter {k} is not contiguous")
            self._handle.set_tensor(k, t)
        self._handle.finalize()
        self._registered = sig

    # ------------------------------------------------------------------ forward
    def forward(self, batch: Dict[str, torch.Tensor], disable_global_head: bool = False,
                disable_local_head: bool = False):
        dev = self._device()
        ctx = self.context()
        coords, feats = batch['coords'], batch['features']
        coords = coords.to(device=dev, dtype=torch.int32).contiguous()
        feats = feats.to(device=dev, dtype=torch.float32).contiguous()
        assert coords.dim() == 2 and coords.shape[1] == 4 and feats.shape == (coords.shape[0], 1)
        bs = batch.get('batch_size', None)
        if bs is None:
            bs = int(coords[:, 0].max().item()) + 1
        ctx.coords_set(coords, bs)
        if self.training:
            return self._forward_train(ctx, feats, disable_global_head, disable_local_head)
        return self._forward_on_plan(ctx, feats, disable_global_head, disable_local_head)

    # eval-mode arithmetic of the sparse convolutions: 'fp32' (exact-fp32 MFMA, the default and what parity is stated
    # for) or 'bf16' (BASELINE configs[2]: MFMA operands rounded to bf16, fp32 accumulate, feature maps fp32)
    precision = 'fp32'

    # process group for SyncBN statistics in train mode (None = this process only); set by the sharded step
    sync_bn_group = None

    def _forward_train(self, ctx: _lib.Context, feats: torch.Tensor, disable_global_head: bool,
                       disable_local_head: bool = False):
        """train mode (reference training/trainer.py:160-175,183-192): batch-statistics BatchNorm, autograd through the
        HIP operators (egonn_amd/train.py).  Output dict as in eval mode; every tensor carries a grad_fn."""
        from . import train
        if not bool((feats == 1).all()):
            raise NotImplementedError("train mode supports the reference's all-ones input features only")
        y = {}
        ctx.prepare_maps(with_level0_transpose=True)          # the tables of all 21 maps of the step in one launch
        levels = train.trunk_forward(self, ctx, self.sync_bn_group)
        if not disable_global_head:
            g = train.global_branch(self, ctx, self.sync_bn_group, levels)
            assert g.dim() == 2 and g.shape[1] == self.global_descriptor_size
            y['global'] = g
        if self.local_head is not None and not disable_local_head:
            lvl, desc, kp_off, sigma = train.local_branch(self, ctx, levels)
            coords = ctx.level_coords(lvl)                                    # (n,4) int32 [b,x,y,z]
            stride = [2 ** lvl] * 3
            if self.ignore_keypoint_regressor:
                kp_off = torch.zeros_like(kp_off)
            kp_pos = self.quantizer.keypoint_position(coords[:, 1:], stride, kp_off)      # minkgl.py:296-302
            off = ctx.level_batch_offsets(lvl)
            B = ctx.batch_size
            y['descriptors'] = [desc[off[b]:off[b + 1]] for b in range(B)]
            y['keypoints'] = [kp_pos[off[b]:off[b + 1]] for b in range(B)]
            y['sigma'] = [sigma[off[b]:off[b + 1]] for b in range(B)]
        return y

    def _forward_on_plan(self, ctx: _lib.Context, feats: torch.Tensor, disable_global_head=False,
                         disable_local_head=False, outputs=None):
        """outputs: (global, descriptors, keypoints, sigma) preallocated tensors for a reserved (capturable) plan —
        the local ones hold `ctx.level_capacity(3)` rows and the call performs no host synchronisation; the returned
        dict then carries the packed tensors instead of per-sample lists."""
        self._sync_weights()
        dev = ctx.device
        B = ctx.batch_size
        lvl = min(LOCAL_LEVELS)
        if outputs is not None:
            return self._forward_reserved(ctx, feats, outputs, disable_global_head, disable_local_head)
        n3 = ctx.level_count(lvl)
        flags = self._flags(disable_global_head, disable_local_head)
        out_g = out_d = out_k = out_s = None
        if not disable_global_head:
            out_g = torch.empty((B, self.global_descriptor_size), dtype=torch.float32, device=dev)
        if not disable_local_head:
            out_d = torch.empty((n3, self.local_descriptor_size), dtype=torch.float32, device=dev)
            out_k = torch.empty((n3, 3), dtype=torch.float32, device=dev)
            out_s = torch.empty((n3, 1), dtype=torch.float32, device=dev)
        q = self.quantizer
        step = (_lib.C.c_float * 3)(*([float(s) for s in q.step] + [0.0, 0.0])[:3])
        with torch.cuda.device(dev):
            _lib.check(ctx.lib.egonn_forward(ctx.h, self._handle.h, _lib._ptr(feats), q.mode, step, flags,
                                             _lib._ptr(out_g), _lib._ptr(out_d), _lib._ptr(out_k), _lib._ptr(out_s),
                                             _lib._stream()))
        y = {}
        if out_g is not None:
            assert out_g.dim() == 2 and out_g.shape[1] == self.global_descriptor_size
            y['global'] = out_g
        if out_d is not None:
            off = ctx.level_batch_offsets(lvl)
            y['descriptors'] = [out_d[off[b]:off[b + 1]] for b in range(B)]
            y['keypoints'] = [out_k[off[b]:off[b + 1]] for b in range(B)]
            y['sigma'] = [out_s[off[b]:off[b + 1]] for b in range(B)]
            self._last_local = (out_d, out_k, out_s)
        return y

    def _flags(self, disable_global_head, disable_local_head):
        flags = 0
        if disable_global_head:
            flags |= _lib.FLAG_DISABLE_GLOBAL
        if disable_local_head:
            flags |= _lib.FLAG_DISABLE_LOCAL
        if self.ignore_keypoint_regressor:
            flags |= _lib.FLAG_IGNORE_KP_REGRESSOR
        if self.precision == 'bf16':
            flags |= _lib.FLAG_BF16
        elif self.precision != 'fp32':
            raise ValueError(f"precision {self.precision!r}: 'fp32' or 'bf16'")
        flags |= {'GeM': 0, 'SPoC': _lib.FLAG_POOL_SPOC, 'MAC': _lib.FLAG_POOL_MAC}[self.global_pool_method]
        return flags

    def _forward_reserved(self, ctx, feats, outputs, disable_global_head=False, disable_local_head=False):
        out_g, out_d, out_k, out_s = outputs
        q = self.quantizer
        step = (_lib.C.c_float * 3)(*([float(s) for s in q.step] + [0.0, 0.0])[:3])
        with torch.cuda.device(ctx.device):
            _lib.check(ctx.lib.egonn_forward(ctx.h, self._handle.h, _lib._ptr(feats), q.mode, step,
                                             self._flags(disable_global_head, disable_local_head),
                                             _lib._ptr(out_g), _lib._ptr(out_d), _lib._ptr(out_k), _lib._ptr(out_s),
                                             _lib._stream()))
        self._last_local = (out_d, out_k, out_s)
        return {'global': out_g, 'descriptors': out_d, 'keypoints': out_k, 'sigma': out_s}

    def keypoint_coords(self) -> List[torch.Tensor]:
        """(n_b,4) int32 super-voxel coordinates of the rows of the last forward's local outputs, per sample
        (the join key for parity checks; ME exposes the same through SparseTensor.C)."""
        ctx = self.context()
        lvl = min(LOCAL_LEVELS)
        c = ctx.level_coords(lvl)
        off = ctx.level_batch_offsets(lvl)
        return [c[off[b]:off[b + 1]] for b in range(ctx.batch_size)]

    def print_info(self):
        # reference models/minkgl.py:317-334
        print(f'Model class: {type(self).__name__}')
        n_params = sum(p.nelement() for p in self.parameters())
        n_trunk = sum(p.nelement() for p in self.trunk.parameters())
        print(f"# parameters - total: {n_params/1000:.1f}   trunk: {n_trunk/1000:.1f} [k]")
        if self.local_head is not None:
            n_local_head = sum(p.nelement() for p in self.local_head.parameters())
            n_kr = sum(p.nelement() for p in self.local_keypoint_regressor.parameters())
            n_kd = sum(p.nelement() for p in self.local_descriptor_decoder.parameters())
            n_sr = sum(p.nelement() for p in self.local_sigma_regressor.parameters())
            print(f'kp. head: {n_local_head/1000:.1f}   kp. regressor {n_kr/1000:.1f}   '
                  f'kp. descriptor {n_kd / 1000:.1f}   [k] kp. saliency regresor {n_sr/1000:.1f} [k]')
            print(f'# channels in the local map: {self.local_head.out_channels}   keypoint descriptor size: '
                  f'{self.local_descriptor_size}')
        n_gh = sum(p.nelement() for p in self.global_head.parameters())
        print(f'global descriptor head: {n_gh/1000:.1f} [k]   pool method: {self.global_pool_method}')
        print(f'# channels in the global map: {self.global_channels}   global descriptor size: '
              f'{self.global_descriptor_size}')


# ----------------------------------------------------------------------------- factory
def create_egonn_model(model_params):
    """reference models/model_factory.py:31-76"""
    if model_params.model != 'egonn':
        raise NotImplementedError(f'Unknown model: {model_params.model}')
    global_in_channels = [PLANES[i - 1] for i in GLOBAL_LEVELS]
    head_global = MinkHead(GLOBAL_LEVELS, global_in_channels, GLOBAL_CH)
    local_in_channels = [PLANES[i - 1] for i in LOCAL_LEVELS]
    head_local = MinkHead(LOCAL_LEVELS, local_in_channels, LOCAL_CH)
    trunk = MinkTrunk(in_channels=1, planes=PLANES, conv0_kernel_size=5)
    return MinkGL(trunk, local_head=head_local, local_descriptor_size=LOCAL_DIM, local_normalize=True,
                  global_head=head_global, global_descriptor_size=GLOBAL_DIM, global_pool_method='GeM',
                  global_normalize=False, quantizer=model_params.quantizer)


def model_factory(model_params):
    """reference models/model_factory.py:12-28"""
    if model_params.model == 'MinkLoc':
        from .minkloc import MinkLoc
        return MinkLoc(in_channels=1, feature_size=model_params.feature_size, output_dim=model_params.output_dim,
                       planes=model_params.planes, layers=model_params.layers,
                       num_top_down=model_params.num_top_down, conv0_kernel_size=model_params.conv0_kernel_size,
                       block=model_params.block, pooling_method=model_params.pooling)
    if model_params.model == 'MinkLoc3D':
        from .minkloc import MinkLoc3D
        return MinkLoc3D()
    if 'egonn' in model_params.model:
        return create_egonn_model(model_params)
    raise NotImplementedError('Model not implemented: {}'.format(model_params.model))
