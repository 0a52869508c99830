"""Seeded synthetic inputs for benchmarks and parity tests (no datasets or pretrained weights
exist in this environment — SURVEY.md §0).

* `lidar_scan`  — the LiDAR-like generator SURVEY.md §8(d) specifies for configs C1/C2
  (64 beams x 2048 azimuth steps, elevation -24.8..+2.0 deg, sensor height 1.73 m, ground
  plane + 60 random axis-aligned boxes, range < 80 m, sigma = 0.02 m noise, subsampled to an
  exact point count).  It must NOT be replaced by the uniform-box generator of the
  reference's `datasets/quantization.py:107-111`: at 0.1 m that yields isolated voxels.
* `seeded_state_dict` — deterministic weights for a given key/shape table, independent of
  any reference code, so fixtures only need to store (seed, outputs).
"""
from __future__ import annotations

import zlib
from typing import Dict, Tuple

import numpy as np


def lidar_scan(seed: int, n_points: int = 50_000, n_beams: int = 64, n_azimuth: int = 2048,
               max_range: float = 80.0, sensor_height: float = 1.73, n_boxes: int = 60,
               noise_sigma: float = 0.02) -> np.ndarray:
    """(n_points, 3) float32 point cloud in the sensor frame (z up, ground at z = -sensor_height)."""
    rng = np.random.default_rng(seed)
    elev = np.deg2rad(np.linspace(-24.8, 2.0, n_beams))
    azim = np.linspace(-np.pi, np.pi, n_azimuth, endpoint=False)
    el, az = np.meshgrid(elev, azim, indexing="ij")
    d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], axis=-1).reshape(-1, 3)

    t = np.full(d.shape[0], np.inf)
    # ground plane z = -sensor_height
    down = d[:, 2] < -1e-6
    t[down] = -sensor_height / d[down, 2]

    # random axis-aligned boxes standing on the ground
    centres = rng.uniform(-70.0, 70.0, size=(n_boxes, 2))
    sizes_xy = rng.uniform(2.0, 25.0, size=(n_boxes, 2))
    heights = rng.uniform(1.5, 15.0, size=n_boxes)
    for c, s, h in zip(centres, sizes_xy, heights):
        lo = np.array([c[0] - s[0] / 2, c[1] - s[1] / 2, -sensor_height])
        hi = np.array([c[0] + s[0] / 2, c[1] + s[1] / 2, -sensor_height + h])
        if lo[0] <= 0.0 <= hi[0] and lo[1] <= 0.0 <= hi[1]:
            continue                                   # box covers the sensor: skipped
        with np.errstate(divide="ignore", invalid="ignore"):
            t0 = lo / d
            t1 = hi / d
        tmin = np.nanmax(np.minimum(t0, t1), axis=1)
        tmax = np.nanmin(np.maximum(t0, t1), axis=1)
        hit = (tmax >= tmin) & (tmin > 0.0)
        t = np.where(hit & (tmin < t), tmin, t)

    keep = np.isfinite(t) & (t < max_range)
    pts = d[keep] * t[keep, None]
    pts = pts + rng.normal(0.0, noise_sigma, size=pts.shape)
    if pts.shape[0] >= n_points:
        sel = rng.choice(pts.shape[0], size=n_points, replace=False)
    else:                                              # rare (tiny configs): sample with replacement + jitter
        sel = rng.choice(pts.shape[0], size=n_points, replace=True)
        pts = pts + 0.0
    pts = pts[sel]
    return np.ascontiguousarray(pts, dtype=np.float32)


def _key_seed(seed: int, key: str) -> int:
    return (int(seed) * 1_000_003 + zlib.crc32(key.encode())) & 0x7FFFFFFF


def seeded_tensor(seed: int, key: str, shape: Tuple[int, ...]) -> np.ndarray:
    """Deterministic fp32 tensor for a state_dict entry.  Distribution by key suffix:
    conv kernels / linear weights ~ N(0, sqrt(2/fan)), BN weight ~ U(0.5,1.5), BN bias and
    running_mean ~ N(0,0.1), running_var ~ U(0.5,1.5), linear bias ~ N(0,0.05), GeM p = 3."""
    rng = np.random.default_rng(_key_seed(seed, key))
    shape = tuple(int(s) for s in shape)
    if key.endswith("num_batches_tracked"):
        return np.zeros(shape, dtype=np.int64)
    if key.endswith("pooling.p"):
        return np.full(shape, 3.0, dtype=np.float32)
    if key.endswith(".kernel"):
        if len(shape) == 3:
            fan = shape[0] * shape[1]                  # K * Cin
        else:
            fan = shape[0]
        return (rng.standard_normal(shape) * np.sqrt(2.0 / fan)).astype(np.float32)
    if key.endswith("eca.conv.weight"):
        return rng.uniform(-0.8, 0.8, size=shape).astype(np.float32)
    if key.endswith("running_var"):
        return rng.uniform(0.5, 1.5, size=shape).astype(np.float32)
    if key.endswith("running_mean"):
        return (rng.standard_normal(shape) * 0.1).astype(np.float32)
    if key.endswith("bn.weight"):
        return rng.uniform(0.5, 1.5, size=shape).astype(np.float32)
    if key.endswith("bn.bias"):
        return (rng.standard_normal(shape) * 0.1).astype(np.float32)
    if key.endswith("linear.weight"):
        return (rng.standard_normal(shape) * np.sqrt(2.0 / shape[1])).astype(np.float32)
    if key.endswith("linear.bias"):
        return (rng.standard_normal(shape) * 0.05).astype(np.float32)
    raise KeyError(f"no seeded distribution for state_dict key {key!r}")


def seeded_state_dict(seed: int, shapes: Dict[str, Tuple[int, ...]]) -> Dict[str, np.ndarray]:
    return {k: seeded_tensor(seed, k, s) for k, s in shapes.items()}
