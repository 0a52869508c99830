"""On-device voxelisers with the reference's Quantizer interface (datasets/quantization.py:8-103).

`quantizer(pc)` -> (coords (m,3) int32, index (m,) int64) exactly like the reference, but computed by
libegonn_hip (floor -> Z-order key -> radix sort -> unique) instead of ME.utils.sparse_quantize.  Output
rows are in Z-order of the voxel coordinate (ME's order is unspecified); `index` is the FIRST point of every
voxel in the original order.  Tensors come back on the device of `pc` (a CPU input is staged through the
GPU, so the reference's CPU-side call sites keep working).
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import List, Sequence

import numpy as np
import torch

from . import _lib


class Quantizer(ABC):
    mode: int
    step: Sequence[float]

    def __init__(self):
        self._ctx = None

    def _context(self, coord_bits: int = 16):
        if self._ctx is None:
            self._ctx = _lib.Context(coord_bits=coord_bits)
        return self._ctx

    def __call__(self, pc):
        """reference datasets/quantization.py:29-44 / :79-85"""
        pc_t = pc if isinstance(pc, torch.Tensor) else torch.as_tensor(np.asarray(pc))
        assert pc_t.dim() == 2 and pc_t.shape[1] == 3
        dev_in = pc_t.device
        ctx = self._context()
        x = pc_t.to(device=ctx.device, dtype=torch.float32).contiguous()
        ctx.voxelize(x, [0, x.shape[0]], self.mode, self.step)
        coords = ctx.level_coords(0)[:, 1:].contiguous()
        index = ctx.input_index()
        return coords.to(dev_in), index.to(dev_in)

    @abstractmethod
    def dequantize(self, coords):
        pass

    @abstractmethod
    def keypoint_position(self, supervoxel_centers, stride, kp_offset):
        pass


class CartesianQuantizer(Quantizer):
    """reference datasets/quantization.py:75-103"""

    def __init__(self, quant_step: float):
        super().__init__()
        self.quant_step = quant_step
        self.mode = _lib.QUANT_CARTESIAN
        self.step = [float(quant_step)]

    def dequantize(self, coords):
        return (0.5 + coords) * self.quant_step

    def keypoint_position(self, supervoxel_centers, stride, kp_offset):
        c = (supervoxel_centers + 0.5) * self.quant_step
        size = torch.tensor(stride, dtype=torch.float, device=c.device) * self.quant_step
        if kp_offset is not None:
            return c + kp_offset * size / 2.
        return c


class PolarQuantizer(Quantizer):
    """reference datasets/quantization.py:22-72"""

    def __init__(self, quant_step: List[float]):
        super().__init__()
        assert len(quant_step) == 3, \
            '3 quantization steps expected: for sector (in degrees), ring and z-coordinate (in meters)'
        self.quant_step = torch.tensor(quant_step, dtype=torch.float)
        self.theta_range = int(360. // self.quant_step[0])
        self.mode = _lib.QUANT_POLAR
        self.step = [float(s) for s in quant_step]

    def to_cartesian(self, pc):
        theta = np.pi * (pc[:, 0] - 180.) / 180.
        x = torch.cos(theta) * pc[:, 1]
        y = torch.sin(theta) * pc[:, 1]
        return torch.stack([x, y, pc[:, 2]], dim=1)

    def dequantize(self, coords):
        return self.to_cartesian((0.5 + coords) * self.quant_step.to(coords.device))

    def keypoint_position(self, supervoxel_centres, stride, kp_offset):
        device = supervoxel_centres.device
        c = (supervoxel_centres + 0.5) * self.quant_step.to(device)
        size = torch.tensor(stride, dtype=torch.float, device=device) * self.quant_step.to(device)
        return self.to_cartesian(c + kp_offset * size / 2.)
