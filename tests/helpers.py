"""Shared helpers for parity tests: golden-fixture loading and coordinate-keyed joins.

Row order is implementation-defined on every side (ME: hash order; oracle: first-appearance;
HIP: Z-order), so every per-row comparison joins on the (b,x,y,z) coordinate first
(SURVEY.md §8c parity protocol)."""
from __future__ import annotations

import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["egonn_cart01_b1", "egonn_cart01_b2", "egonn_cart03_b1", "egonn_polar_b1",
         "egonn_cart01_50k_b2"]      # the last one: BASELINE configs[1] cloud size (2 x 50 000 points, 0.1 m)


def load_case(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


MINKLOC_CASES = ["minkloc3d_cart03_b2", "minkloc_eca_cart03"]


def state_dict_shapes(name="egonn"):
    with open(os.path.join(GOLDEN, f"{name}_state_dict_shapes.json")) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}        # dict order = reference state_dict order


def seeded_weights(seed, name="egonn"):
    from egonn_amd.synth import seeded_state_dict
    return seeded_state_dict(int(seed), state_dict_shapes(name))


def rowkey(c4):
    c = np.asarray(c4, dtype=np.int64)
    return ((c[:, 0] * 65536 + (c[:, 1] + 32768)) * 65536 + (c[:, 2] + 32768)) * 65536 + (c[:, 3] + 32768)


def sort_rows(c4):
    c = np.asarray(c4)
    return c[np.argsort(rowkey(c), kind="stable")]


def join_perm(c_from, c_to):
    """perm such that c_from[perm] == c_to row by row (asserts identical coordinate sets)."""
    kf, kt = rowkey(c_from), rowkey(c_to)
    assert len(kf) == len(kt), f"row count differs: {len(kf)} vs {len(kt)}"
    of = np.argsort(kf, kind="stable")
    ot = np.argsort(kt, kind="stable")
    assert np.array_equal(kf[of], kt[ot]), "coordinate sets differ"
    perm = np.empty(len(kt), dtype=np.int64)
    perm[ot] = of
    return perm


def cosine_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    num = (a * b).sum(axis=1)
    den = np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1)
    return 1.0 - num / np.maximum(den, 1e-30)


def make_quantizer(case, mod):
    """Build the matching quantiser from module `mod` (oracle.egonn_ref or egonn_amd)."""
    step = case["quantization_step"]
    if str(case["coordinates"]) == "cartesian":
        return mod.CartesianQuantizer(float(step[0]))
    return mod.PolarQuantizer([float(s) for s in step])
