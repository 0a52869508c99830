"""MinkLoc / MinkLoc3D (MinkFPN backbone + GeM) with the reference's Python surface, executed through the
per-operator entry points of libegonn_hip (reference: models/minkfpn.py, models/minkloc.py,
third_party/minkloc3d/minkloc.py, models/resnet.py:81-117).  Same kernels as EgoNN, second graph; the module
tree only holds parameters (identical state_dict keys/shapes), there is no PyTorch fallback.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.nn as nn

from . import _lib
from .model import SparseConv, BatchNorm, ECALayer, PoolingWrapper, GeM


class BasicBlock(nn.Module):
    """ME modules.resnet_block.BasicBlock parameters (conv1 norm1 conv2 norm2 [downsample])."""
    expansion = 1

    def __init__(self, inplanes, planes, downsample=None, eca: bool = False):
        super().__init__()
        self.conv1 = SparseConv(inplanes, planes, 3)
        self.norm1 = BatchNorm(planes)
        self.conv2 = SparseConv(planes, planes, 3)
        self.norm2 = BatchNorm(planes)
        self.downsample = downsample
        if eca:
            self.eca = ECALayer(planes, gamma=2, b=1)      # reference layers/eca_block.py:54


class MinkFPN(nn.Module):
    """reference models/minkfpn.py:9-93 (parameter layout of network_initialization :26-63)."""

    def __init__(self, in_channels, out_channels, num_top_down=1, conv0_kernel_size=5, block='BasicBlock',
                 layers: Sequence[int] = (1, 1, 1), planes: Sequence[int] = (32, 64, 64)):
        super().__init__()
        assert len(layers) == len(planes) and 1 <= len(layers) and 0 <= num_top_down <= len(layers)
        if block not in ('BasicBlock', 'ECABasicBlock'):
            raise NotImplementedError(f'block {block!r}: the MI355X path implements BasicBlock and ECABasicBlock')
        self.num_bottom_up, self.num_top_down = len(layers), num_top_down
        self.layers, self.planes, self.lateral_dim = list(layers), list(planes), out_channels
        eca = block == 'ECABasicBlock'
        self.convs, self.bn, self.blocks = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.tconvs, self.conv1x1 = nn.ModuleList(), nn.ModuleList()
        inplanes = planes[0]
        self.conv0 = SparseConv(in_channels, inplanes, conv0_kernel_size)
        self.bn0 = BatchNorm(inplanes)
        for plane, layer in zip(planes, layers):
            self.convs.append(SparseConv(inplanes, inplanes, 2))
            self.bn.append(BatchNorm(inplanes))
            down = None
            if inplanes != plane:
                down = nn.Sequential(SparseConv(inplanes, plane, 1), BatchNorm(plane))
            blocks = [BasicBlock(inplanes, plane, down, eca)]
            inplanes = plane
            blocks += [BasicBlock(inplanes, plane, None, eca) for _ in range(1, layer)]
            self.blocks.append(nn.Sequential(*blocks))
        for i in range(num_top_down):
            self.conv1x1.append(SparseConv(planes[-1 - i], out_channels, 1))
            self.tconvs.append(SparseConv(out_channels, out_channels, 2, transpose=True))
        if num_top_down < self.num_bottom_up:
            self.conv1x1.append(SparseConv(planes[-1 - num_top_down], out_channels, 1))
        else:
            self.conv1x1.append(SparseConv(planes[0], out_channels, 1))

    # ------------------------------------------------------------------ forward on a plan (minkfpn.py:65-93)
    def run(self, ctx: _lib.Context, feats0: torch.Tensor):
        fold = ctx.bn_fold

        def conv_bn(lin, lout, k, x, conv, bn, relu):
            sc, sh = fold(bn.bn)
            return ctx.conv(lin, lout, k, x, conv.kernel.detach(), sc, sh, relu=relu)

        def block(level, x, b: BasicBlock):
            t = conv_bn(level, level, 3, x, b.conv1, b.norm1, True)
            t = conv_bn(level, level, 3, t, b.conv2, b.norm2, False)
            res = x if b.downsample is None else conv_bn(level, level, 1, x, b.downsample[0], b.downsample[1], False)
            return ctx.block_tail(level, t, res, b.eca.conv.weight if hasattr(b, 'eca') else None)

        x = conv_bn(0, 0, self.conv0.kernel_size, feats0, self.conv0, self.bn0, True)
        fmaps = []
        if self.num_top_down == self.num_bottom_up:
            fmaps.append((0, x))
        level = 0
        for ndx, (conv, bn, blocks) in enumerate(zip(self.convs, self.bn, self.blocks)):
            x = conv_bn(level, level + 1, 2, x, conv, bn, True)
            level += 1
            for b in blocks:
                x = block(level, x, b)
            if self.num_bottom_up - 1 - self.num_top_down <= ndx < len(self.convs) - 1:
                fmaps.append((level, x))
        assert len(fmaps) == self.num_top_down
        x = ctx.conv(level, level, 1, x, self.conv1x1[0].kernel.detach())
        for ndx, tconv in enumerate(self.tconvs):
            x = ctx.conv_transpose(level, x, tconv.kernel.detach())
            level -= 1
            flevel, f = fmaps[-ndx - 1]
            assert flevel == level
            x = ctx.add(x, ctx.conv(level, level, 1, f, self.conv1x1[ndx + 1].kernel.detach()))
        return level, x


class _MinkLocBase(nn.Module):
    coord_bits = 16

    def _device(self):
        dev = next(self.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError("egonn_amd MinkLoc models run on MI355X only: move the model to a HIP device "
                               "(`model.to('cuda')`); there is no CPU fallback.")
        return dev

    def context(self) -> _lib.Context:
        dev = self._device()
        if getattr(self, '_ctx', None) is None or self._ctx.device != dev:
            self._ctx = _lib.Context(dev, coord_bits=self.coord_bits)
        return self._ctx

    sync_bn_group = None          # process group of the SyncBN statistics in train mode (None = this process)

    def _forward(self, batch: Dict[str, torch.Tensor], gem_p: torch.Tensor):
        dev = self._device()
        ctx = self.context()
        coords = batch['coords'].to(device=dev, dtype=torch.int32).contiguous()
        feats = batch['features'].to(device=dev, dtype=torch.float32).contiguous()
        assert coords.dim() == 2 and coords.shape[1] == 4 and feats.shape[0] == coords.shape[0]
        bs = batch.get('batch_size', None)
        if bs is None:
            bs = int(coords[:, 0].max().item()) + 1
        ctx.coords_set(coords, bs)
        if self.training:     # batch-statistics BatchNorm + autograd through the HIP operators (egonn_amd/train.py)
            from . import train
            if not bool((feats == 1).all()):
                raise NotImplementedError("train mode supports the reference's all-ones input features only")
            level, x = train.minkfpn_forward(self.backbone, ctx, self.sync_bn_group)
            assert x.shape[1] == self.feature_size
            return {'global': train.GeMFn.apply(x, gem_p, ctx, level)}
        with torch.no_grad():
            level, x = self.backbone.run(ctx, ctx.gather_input(feats))
            assert x.shape[1] == self.feature_size
            g = ctx.gem(level, x, gem_p)
        assert g.dim() == 2 and g.shape[1] == self.output_dim
        return {'global': g}


class MinkLoc(_MinkLocBase):
    """reference models/minkloc.py:13-75"""

    def __init__(self, in_channels, feature_size, output_dim, planes, layers, num_top_down, conv0_kernel_size,
                 block='BasicBlock', pooling_method='GeM'):
        super().__init__()
        self.in_channels, self.feature_size, self.output_dim, self.block = in_channels, feature_size, output_dim, block
        self.pooling_method = pooling_method
        self.backbone = MinkFPN(in_channels=in_channels, out_channels=feature_size, num_top_down=num_top_down,
                                conv0_kernel_size=conv0_kernel_size, block=block, layers=layers, planes=planes)
        self.pooling = PoolingWrapper(pool_method=pooling_method, in_dim=feature_size, output_dim=output_dim)
        self.pooled_feature_size = self.pooling.output_dim

    def forward(self, batch):
        return self._forward(batch, self.pooling.pooling.p)

    def print_info(self):
        print('Model class: MinkLoc')
        print('Total parameters: {}'.format(sum(p.nelement() for p in self.parameters())))
        print('Backbone parameters: {}'.format(sum(p.nelement() for p in self.backbone.parameters())))
        print('Backbone building block: {}'.format(self.block))
        print('Pooling method: {}'.format(self.pooling_method))


class MinkLoc3D(_MinkLocBase):
    """reference third_party/minkloc3d/minkloc.py:9-44"""

    def __init__(self):
        super().__init__()
        self.feature_size = self.output_dim = 256
        self.backbone = MinkFPN(in_channels=1, out_channels=256, num_top_down=1, conv0_kernel_size=5,
                                layers=[1, 1, 1], planes=[32, 64, 64])
        self.pooling = GeM(input_dim=256)

    def forward(self, batch, disable_local_head: bool = True):
        assert disable_local_head, "MinkLoc3D model has only the global head"
        return self._forward(batch, self.pooling.p)

    def print_info(self):
        print('Model class: MinkLoc')
        print('Total parameters: {}'.format(sum(p.nelement() for p in self.parameters())))
