"""Scan ingest: raw LiDAR files -> filtered device points, the step in front of the descriptor path.

Mirrors the reference loaders (datasets/mulran/mulran_raw.py:14-25, datasets/kitti/kitti_raw.py:11-22,
misc/point_clouds.py:78-111): `.bin` = float32 (x, y, z, reflectance) per return; all-zero returns and returns at or
below the dataset's ground-plane level are dropped.  Here the raw payload of a whole batch goes to the GPU in ONE
pinned-host -> device copy and is filtered/compacted there (libegonn_hip `egonn_filter_points`), so that the host only
reads files."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _lib

GROUND_PLANE_LEVEL = {"mulran": -0.9, "kitti": -1.5, "southbay": -1.6}


def read_bin(path: str) -> np.ndarray:
    """(n, 4) float32 view of a MulRan / KITTI `.bin` scan (no preprocessing)."""
    return np.fromfile(path, dtype=np.float32).reshape(-1, 4)


class ScanIngest:
    def __init__(self, dataset_type: str = "mulran", device=None, remove_zero_points: bool = True,
                 remove_ground_plane: bool = True):
        self.ground = float(GROUND_PLANE_LEVEL[dataset_type])
        self.remove_zero_points, self.remove_ground_plane = remove_zero_points, remove_ground_plane
        self.device = device if device is not None else _lib.require_gpu()
        self._pinned = None

    def _stage(self, raws: Sequence[np.ndarray]) -> Tuple[torch.Tensor, List[int], int]:
        stride = raws[0].shape[1]
        assert stride in (3, 4) and all(r.ndim == 2 and r.shape[1] == stride and r.dtype == np.float32 for r in raws)
        off = [0]
        for r in raws:
            off.append(off[-1] + len(r))
        n = off[-1]
        if self._pinned is None or self._pinned.numel() < n * stride:
            self._pinned = torch.empty(max(n * stride, 1), dtype=torch.float32).pin_memory()
        host = self._pinned[: n * stride].view(n, stride)
        for r, lo, hi in zip(raws, off[:-1], off[1:]):
            host[lo:hi] = torch.from_numpy(r)
        return host, off, stride

    def __call__(self, raws: Sequence[np.ndarray]) -> Tuple[torch.Tensor, List[int]]:
        """raw scans (n_b, 4|3) float32 -> (points (N,3) on the device, per-scan offsets of the survivors)."""
        host, off, stride = self._stage(raws)
        return self._filter(host, off, stride)

    def _filter(self, host: torch.Tensor, off: List[int], stride: int) -> Tuple[torch.Tensor, List[int]]:
        lib = _lib.load()
        n, B = off[-1], len(off) - 1
        dev = self.device
        raw = host.to(dev, non_blocking=True)
        raw_off = torch.tensor(off, dtype=torch.int64).to(dev, non_blocking=True)
        out = torch.empty((max(n, 1), 3), dtype=torch.float32, device=dev)
        new_off = torch.empty(B + 1, dtype=torch.int64, device=dev)
        scratch = torch.empty(lib.egonn_filter_points_scratch_ints(n), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.egonn_filter_points(raw.data_ptr(), n, stride, raw_off.data_ptr(), B,
                                               int(self.remove_zero_points), int(self.remove_ground_plane), self.ground,
                                               out.data_ptr(), new_off.data_ptr(), scratch.data_ptr(), scratch.numel(),
                                               _lib._stream()))
        offs = new_off.tolist()                                   # the one host sync of the ingest step (B+1 values)
        return out[: offs[-1]], offs

    def load(self, paths: Sequence[str]) -> Tuple[torch.Tensor, List[int]]:
        """`.bin` files -> device points: every file is read straight into the pinned staging buffer (no intermediate
        array), then one H2D copy + the device filter."""
        import os
        sizes = [os.path.getsize(p) for p in paths]
        assert all(sz % 16 == 0 for sz in sizes), "a .bin scan is float32 x,y,z,reflectance records"
        off = [0]
        for sz in sizes:
            off.append(off[-1] + sz // 16)
        n = off[-1]
        if self._pinned is None or self._pinned.numel() < n * 4:
            self._pinned = torch.empty(max(n * 4, 1), dtype=torch.float32).pin_memory()
        buf = memoryview(self._pinned.numpy()).cast("B")
        for p, lo, hi in zip(paths, off[:-1], off[1:]):
            with open(p, "rb", buffering=0) as f:
                got = f.readinto(buf[lo * 16: hi * 16])
                assert got == (hi - lo) * 16, p
        return self._filter(self._pinned[: n * 4].view(n, 4), off, 4)
