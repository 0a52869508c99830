"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
  * the golden vectors produced by the reference's own graph code (tests/golden/*.npz),
  * the CPU oracle (oracle/me_ops.py, oracle/egonn_ref.py) on seeded inputs,
  * size-independent properties at BASELINE.json's full size (50k-pt clouds, batch 16).

Bars: integer voxel coordinates / indices / keypoint selection bit-exact; fp32 descriptors within 1e-4
cosine (BASELINE.json north_star); other fp32 tensors within the rtol/atol written in each test.
Row order differs between implementations, so everything joins on the (b,x,y,z) coordinate."""
import os

import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import __graft_entry__ as g
    g.build()
    import egonn_amd
    return egonn_amd


def _model(gpu, case=None, seed=None, coordinates="cartesian", step=0.1):
    if case is not None:
        coordinates = str(case["coordinates"])
        st = case["quantization_step"]
        step = float(st[0]) if coordinates == "cartesian" else [float(s) for s in st]
        seed = int(case["weight_seed"])
    mp = gpu.ModelParams(model="egonn", coordinates=coordinates, quantization_step=step)
    m = gpu.model_factory(mp)
    w = H.seeded_weights(seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    return m.to("cuda").eval(), w


def _np(t):
    return t.detach().cpu().numpy()


def _gap_ok(sig, tol=2e-4):
    """positions of an ascending sigma list whose value is separated from both neighbours by more than the forward
    tolerance: there the selection order must match exactly (elsewhere fp32 noise may swap near-equal saliencies)."""
    sig = np.asarray(sig, dtype=np.float64).reshape(-1)
    if len(sig) < 2:
        return np.ones(len(sig), bool)
    d = np.diff(sig) > tol * np.maximum(1.0, np.abs(sig[1:]))
    return np.r_[True, d] & np.r_[d, True]


# ------------------------------------------------------------------------------------ voxeliser (a1, a2)
@pytest.mark.parametrize("name", H.CASES)
def test_quantizer_matches_reference(gpu, name):
    case = H.load_case(name)
    m, _ = _model(gpu, case)
    for b in range(int(case["n_scans"])):
        pts = torch.from_numpy(case[f"points_{b}"])
        coords, idx = m.quantizer(pts)                       # CPU in -> CPU out, like the reference
        coords, idx = _np(coords), _np(idx)
        assert coords.dtype == np.int32 and idx.dtype == np.int64
        g_c, g_i = case[f"quant_coords_{b}"], case[f"quant_index_{b}"]
        c4 = np.concatenate([np.zeros((len(coords), 1), np.int32), coords], axis=1)
        g4 = np.concatenate([np.zeros((len(g_c), 1), np.int32), g_c], axis=1)
        if str(case["coordinates"]) == "cartesian":          # integer work: bit-exact
            perm = H.join_perm(c4, g4)
            assert np.array_equal(coords[perm], g_c)
            assert np.array_equal(idx[perm], g_i)            # first point of every voxel
        else:                                                # atan2 in fp64 rounded once: the same bins as the reference
            perm = H.join_perm(c4, g4)
            assert np.array_equal(coords[perm], g_c)
            assert np.array_equal(idx[perm], g_i)
        # Z-order output is sorted by construction and duplicate-free
        assert len(np.unique(H.rowkey(c4))) == len(c4)


def test_quantizer_edge_cases(gpu):
    from oracle import me_ops as ops
    q = gpu.CartesianQuantizer(0.1)
    # single point, duplicates, negative coordinates (true floor), exact bin edges
    pts = np.array([[0.05, 0.0, 0.0], [-0.05, 0.0, 0.0], [0.06, 0.01, 0.0], [0.31, 0.0, -0.11],
                    [-0.1, -0.2, -0.3], [0.3, 0.2, 0.1], [-0.1, -0.2, -0.3]], np.float32)
    c, i = q(torch.from_numpy(pts))
    oc, oi = ops.sparse_quantize(pts, 0.1)
    perm = H.join_perm(np.c_[np.zeros(len(c), np.int32), _np(c)], np.c_[np.zeros(len(oc), np.int32), oc])
    assert np.array_equal(_np(c)[perm], oc) and np.array_equal(_np(i)[perm], oi)
    c1, i1 = q(torch.from_numpy(pts[:1]))
    assert _np(c1).tolist() == [[0, 0, 0]] and _np(i1).tolist() == [0]
    # out-of-range coordinates must fail loudly, not wrap
    with pytest.raises(RuntimeError, match="range"):
        q(torch.tensor([[1.0e6, 0.0, 0.0]]))
    with pytest.raises(RuntimeError, match="range"):
        q(torch.tensor([[float("nan"), 0.0, 0.0]]))


def test_quantizer_random_cloud_vs_oracle(gpu):
    from oracle import me_ops as ops
    rng = np.random.default_rng(5)
    pts = rng.uniform(-60, 60, size=(200000, 3)).astype(np.float32)
    pts[:, 2] = rng.uniform(-3, 8, size=len(pts))
    for qs in (0.1, 0.3, 1.0):
        c, i = gpu.CartesianQuantizer(qs)(torch.from_numpy(pts).cuda())
        oc, oi = ops.sparse_quantize(pts, qs)
        perm = H.join_perm(np.c_[np.zeros(len(c), np.int32), _np(c)], np.c_[np.zeros(len(oc), np.int32), oc])
        assert np.array_equal(_np(c)[perm], oc) and np.array_equal(_np(i)[perm], oi)


@pytest.mark.parametrize("cb,n_scans,max_pts", [(13, 7, 3000), (10, 64, 3000), (12, 64, 3000), (15, 64, 3000), (16, 5, 40000),
                                                (11, 1, 5000), (12, 3, 0)])
def test_batched_voxelize_segmented_sort_variants(gpu, cb, n_scans, max_pts):
    """The segmented radix sort of plans built from points (csrc/sort.hip), through the C ABI: odd and even pass counts
    (3 * coord_bits key bits, 9 per pass: 4 / 5 / 6 passes), 64 scans with empty ones in front, in the middle and at the end,
    PACKED elements (Morton bits | index in scan: whenever the point count fits the spare bits) and (key, value) pairs
    (coord_bits 16 with more than 65 536 points), an all-empty-but-one batch.  Per scan: the voxel set and the first-point index
    of every voxel against the reference restatement (datasets/quantization.py:79-85 -> ME sparse_quantize)."""
    from egonn_amd import _lib
    from oracle import me_ops as ops
    dev = _lib.require_gpu()
    rng = np.random.default_rng(1000 * cb + n_scans)
    lim = min(0.1 * ((1 << (cb - 1)) - 2), 60.0)
    sizes = rng.integers(0, max_pts + 1, size=n_scans) if max_pts > 0 else np.zeros(n_scans, np.int64)
    if n_scans >= 7:
        sizes[[0, 3, n_scans - 1]] = 0                      # empty scans in front, inside, at the end
    if max_pts == 0:
        sizes[1] = 257                                      # all scans empty but one
    if cb == 16:
        sizes[:] = max_pts                                  # 200 000 points: beyond the 16 spare bits -> (key, value) pairs
    scans = []
    for n in sizes:
        p = rng.uniform(-lim, lim, size=(int(n), 3)).astype(np.float32)
        p[:, 2] = rng.uniform(-min(lim, 3.0), min(lim, 8.0), size=int(n))
        if n > 10:
            p[5:10] = p[0:5]                                # exact duplicates: the FIRST point of a voxel wins
        scans.append(p)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    ctx = _lib.Context(dev, coord_bits=cb)
    allp = np.concatenate(scans) if off[-1] > 0 else np.zeros((0, 3), np.float32)
    ctx.voxelize(torch.from_numpy(np.ascontiguousarray(allp)).to(dev), off.tolist(), 0, [0.1])
    coords = _np(ctx.level_coords(0))
    first = _np(ctx.input_index())
    boff = ctx.level_batch_offsets(0)
    assert boff[0] == 0 and boff[-1] == len(coords)
    for b in range(n_scans):
        c, f = coords[boff[b]:boff[b + 1]], first[boff[b]:boff[b + 1]]
        oc, oi = (ops.sparse_quantize(scans[b], 0.1) if len(scans[b]) else (np.zeros((0, 3), np.int32), np.zeros((0,), np.int64)))
        assert len(c) == len(oc), (b, len(c), len(oc))
        if len(c) == 0:
            continue
        assert (c[:, 0] == b).all()
        perm = H.join_perm(c, np.c_[np.full(len(oc), b, np.int32), oc])
        assert np.array_equal(c[perm][:, 1:], oc)
        assert np.array_equal(f[perm], oi)                  # first point of every voxel, as an index into its scan


@pytest.mark.parametrize("case", ["dense_120k", "polar_120k", "one_voxel", "outliers", "many_tiny", "two_cells"])
def test_voxelize_sort_stress_inputs(gpu, case):
    """Stress inputs of the segmented sort behind plans built from points (csrc/sort.hip), through the C ABI: a dense 120 k-point
    scan at 0.3 m, polar coordinates (all keys share their top digits), every point in ONE voxel (no varying bit), far outliers next
    to a dense core, 64 scans of 0..300 points, and two occupied cells 200 m apart.  (Written for the MSD-first variant of round 6
    — tools/exp/r06_sort_msd_first.patch: bit-exact on all of these, slower than the four LSD passes, not shipped — and kept for
    the LSD passes.)  Per scan: voxel set and first-point index against datasets/quantization.py:29-44,79-85 as restated in
    oracle/me_ops.py."""
    from egonn_amd import _lib
    from oracle import me_ops as ops
    dev = _lib.require_gpu()
    import zlib
    rng = np.random.default_rng(zlib.crc32(case.encode()) % 1000)    # (fixed data: the numpy polar restatement takes atan2 in fp32 and can
                                                                     #  disagree with the fp64-rounded-once quantiser on a bin edge)
    mode, step, q = 0, [0.1], 0.1
    if case == "dense_120k":
        p = (rng.standard_normal((120000, 3)) * np.array([6.0, 6.0, 0.6])).astype(np.float32)
        scans, q, step = [p, p[:40000] * 0.5], 0.3, [0.3]
    elif case == "polar_120k":
        p = (rng.standard_normal((120000, 3)) * np.array([25.0, 25.0, 1.5])).astype(np.float32)
        scans, mode, step = [p, p[::3]], 1, [1.0, 0.3, 0.2]
    elif case == "one_voxel":
        scans = [(rng.uniform(0.01, 0.09, (5000, 3)) + np.array([3.0, -2.0, 1.0])).astype(np.float32), rng.uniform(-5, 5, (900, 3)).astype(np.float32)]
    elif case == "outliers":
        core = (rng.standard_normal((60000, 3)) * np.array([3.0, 3.0, 0.4])).astype(np.float32)
        far = rng.uniform(-200, 200, (40, 3)).astype(np.float32)
        scans = [np.concatenate([far[:20], core, far[20:]])]
    elif case == "many_tiny":
        scans = [rng.uniform(-40, 40, (int(n), 3)).astype(np.float32) for n in rng.integers(0, 301, size=64)]
    else:
        a = rng.uniform(-2, 2, (9000, 3)).astype(np.float32)
        scans = [np.concatenate([a + np.array([150.0, 150.0, 3.0], np.float32), a - np.array([150.0, 150.0, 3.0], np.float32)])]
    sizes = [len(x) for x in scans]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    ctx = _lib.Context(dev, coord_bits=12)
    ctx.voxelize(torch.from_numpy(np.ascontiguousarray(np.concatenate(scans))).to(dev), off.tolist(), mode, step)
    coords = _np(ctx.level_coords(0))
    first = _np(ctx.input_index())
    boff = ctx.level_batch_offsets(0)
    for b in range(len(scans)):
        c, f = coords[boff[b]:boff[b + 1]], first[boff[b]:boff[b + 1]]
        if len(scans[b]) == 0:
            assert len(c) == 0
            continue
        if mode == 0:
            oc, oi = ops.sparse_quantize(scans[b], q)
        else:
            from oracle import egonn_ref as ref
            oc, oi = ref.PolarQuantizer(step)(scans[b])
            oc, oi = np.asarray(oc), np.asarray(oi)
        assert len(c) == len(oc), (case, b, len(c), len(oc))
        perm = H.join_perm(c, np.c_[np.full(len(oc), b, np.int32), oc])
        assert np.array_equal(c[perm][:, 1:], oc)
        assert np.array_equal(f[perm], oi)


# ------------------------------------------------------------------------------------ coordinate pyramid (a3)
@pytest.mark.parametrize("name", H.CASES)
def test_pyramid_matches_reference(gpu, name):
    from oracle import me_ops as ops
    case = H.load_case(name)
    ctx = gpu._lib.Context()
    c4 = case["coords"]
    rng = np.random.default_rng(0)
    shuffled = c4[rng.permutation(len(c4))]                   # "must not assume sorted input"
    ctx.coords_set(torch.from_numpy(shuffled).cuda(), int(case["n_scans"]))
    assert np.array_equal(H.sort_rows(_np(ctx.level_coords(0))), H.sort_rows(c4))
    for lvl in range(1, 8):
        got = _np(ctx.level_coords(lvl))
        assert len(np.unique(H.rowkey(got))) == len(got)
        if lvl >= 3:
            assert np.array_equal(H.sort_rows(got), case[f"level{lvl}_coords"])
        else:
            assert np.array_equal(H.sort_rows(got), H.sort_rows(ops.stride_coords(c4, 1 << lvl)))
    # input_index: row i of level 0 came from caller row index[i]
    idx = _np(ctx.input_index())
    assert np.array_equal(shuffled[idx], _np(ctx.level_coords(0)))
    # per-sample offsets
    off = ctx.level_batch_offsets(3)
    b = _np(ctx.level_coords(3))[:, 0]
    assert off[0] == 0 and off[-1] == len(b)
    for s in range(int(case["n_scans"])):
        assert (b[off[s]:off[s + 1]] == s).all()


def test_coords_duplicates_and_errors(gpu):
    ctx = gpu._lib.Context()
    c = torch.tensor([[0, 1, 2, 3], [0, -5, 0, 7], [0, 1, 2, 3], [1, 0, 0, 0]], dtype=torch.int32).cuda()
    ctx.coords_set(c, 2)
    assert ctx.level_count(0) == 3                             # duplicate collapsed onto its first occurrence
    idx = _np(ctx.input_index())
    assert sorted(idx.tolist()) == [0, 1, 3]
    with pytest.raises(RuntimeError, match="batch"):
        ctx.coords_set(c, 1)                                   # batch index 1 >= batch size 1
    big = torch.tensor([[0, 40000, 0, 0]], dtype=torch.int32).cuda()
    with pytest.raises(RuntimeError, match="range"):
        ctx.coords_set(big, 1)
    small = gpu._lib.Context(coord_bits=10)                    # +-512 voxels
    with pytest.raises(RuntimeError, match="range"):
        small.coords_set(torch.tensor([[0, 600, 0, 0]], dtype=torch.int32).cuda(), 1)
    small.coords_set(torch.tensor([[0, 511, -512, 0]], dtype=torch.int32).cuda(), 1)
    assert _np(small.level_coords(0)).tolist() == [[0, 511, -512, 0]]
    assert _np(small.level_coords(7)).tolist() == [[0, 384, -512, 0]]


# ------------------------------------------------------------------------------------ sparse-conv primitives (a4-a6)
def _oracle_levels(c4):
    from oracle import egonn_ref as ref
    return ref.SparseLevels(c4)


@pytest.mark.parametrize("naive", [False, True])
def test_conv_primitives_vs_oracle(gpu, naive):
    from oracle import me_ops as ops
    case = H.load_case("egonn_cart01_b2")
    c4 = case["coords"]
    lv = _oracle_levels(c4)
    ctx = gpu._lib.Context()
    ctx.coords_set(torch.from_numpy(c4).cuda(), 2)
    rng = np.random.default_rng(1)
    ctx.set_naive_conv(naive)
    try:
        def feats(level, c):
            """random features defined per coordinate; returns (oracle-order, hip-order) arrays"""
            f = rng.standard_normal((lv.n(level), c)).astype(np.float32)
            perm = H.join_perm(lv.coords[level], _np(ctx.level_coords(level)))
            return f, f[perm]

        def check(level, got, want, tol=2e-5):
            perm = H.join_perm(_np(ctx.level_coords(level)), lv.coords[level])
            got = _np(got)[perm]
            scale = np.abs(want).max() + 1e-6
            assert np.abs(got - want).max() / scale < tol, (level, np.abs(got - want).max() / scale)

        # k=3 at several levels / channel plans (all kernel instantiations of the EgoNN trunk)
        for level, cin, cout in [(1, 32, 32), (2, 32, 64), (2, 64, 64), (4, 64, 128), (4, 128, 128), (6, 128, 128)]:
            fo, fh = feats(level, cin)
            w = (rng.standard_normal((27, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
            sc = rng.uniform(0.5, 1.5, cout).astype(np.float32)
            sh = rng.standard_normal(cout).astype(np.float32) * 0.1
            want = ops.conv_forward(fo, w, lv.kmap(level, level, 3), lv.n(level))
            got = ctx.conv(level, level, 3, torch.from_numpy(fh), torch.from_numpy(w))
            check(level, got, want)
            got = ctx.conv(level, level, 3, torch.from_numpy(fh), torch.from_numpy(w), torch.from_numpy(sc),
                           torch.from_numpy(sh), relu=True)
            check(level, got, np.maximum(want * sc + sh, 0))
        # k=2, s=2
        for level, c in [(0, 32), (1, 32), (2, 64), (4, 128)]:
            fo, fh = feats(level, c)
            w = (rng.standard_normal((8, c, c)) / np.sqrt(2 * c)).astype(np.float32)
            want = ops.conv_forward(fo, w, lv.kmap(level, level + 1, 2), lv.n(level + 1))
            got = ctx.conv(level, level + 1, 2, torch.from_numpy(fh), torch.from_numpy(w))
            check(level + 1, got, want)
        # transposed k=2, s=2 onto the cached finer map
        for level, c in [(4, 64), (7, 128), (6, 128)]:
            fo, fh = feats(level, c)
            w = (rng.standard_normal((8, c, c)) / np.sqrt(c)).astype(np.float32)
            want = ops.conv_transpose_forward(fo, w, lv.kmap(level - 1, level, 2), lv.n(level - 1))
            got = ctx.conv_transpose(level, torch.from_numpy(fh), torch.from_numpy(w))
            check(level - 1, got, want)
        # k=5, Cin=1 (non-constant features to exercise the general path)
        fo, fh = feats(0, 1)
        w = (rng.standard_normal((125, 1, 32)) / np.sqrt(20)).astype(np.float32)
        want = ops.conv_forward(fo, w, lv.kmap(0, 0, 5), lv.n(0))
        got = ctx.conv(0, 0, 5, torch.from_numpy(fh), torch.from_numpy(w))
        check(0, got, want)
        # 1x1
        fo, fh = feats(3, 64)
        w = (rng.standard_normal((64, 128)) / 8).astype(np.float32)
        check(3, ctx.conv(3, 3, 1, torch.from_numpy(fh), torch.from_numpy(w)), fo @ w)
        # global average pooling
        fo, fh = feats(2, 64)
        got = _np(ctx.global_avg_pool(2, torch.from_numpy(fh)))
        np.testing.assert_allclose(got, ops.global_avg_pool(fo, lv.coords[2], 2), rtol=1e-4, atol=1e-5)
    finally:
        ctx.set_naive_conv(False)


def test_conv_every_channel_pair_at_split_levels(gpu):
    """generic operator API: every Cin, Cout in {32, 64, 128} at a level where the product choice is the split-bf16 /
    window kernel family — pairs without a split instantiation (32->128, 128->32) must fall through to the exact
    kernels (they failed with 'no instantiation')."""
    from oracle import me_ops as ops
    case = H.load_case("egonn_cart01_b2")
    c4 = case["coords"]
    lv = _oracle_levels(c4)
    ctx = gpu._lib.Context()
    ctx.coords_set(torch.from_numpy(c4).cuda(), 2)
    rng = np.random.default_rng(5)
    for level in (2, 4):
        perm_in = H.join_perm(lv.coords[level], _np(ctx.level_coords(level)))
        perm_out = H.join_perm(_np(ctx.level_coords(level)), lv.coords[level])
        for cin in (32, 64, 128):
            for cout in (32, 64, 128):
                f = rng.standard_normal((lv.n(level), cin)).astype(np.float32)
                w = (rng.standard_normal((27, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
                want = ops.conv_forward(f, w, lv.kmap(level, level, 3), lv.n(level))
                got = _np(ctx.conv(level, level, 3, torch.from_numpy(f[perm_in]), torch.from_numpy(w)))[perm_out]
                err = np.abs(got - want).max() / (np.abs(want).max() + 1e-6)
                assert err < 2e-5, (level, cin, cout, err)


def test_conv_known_answer_line(gpu):
    """hand-derivable case: three voxels on a line (same as tests/test_oracle.py)."""
    ctx = gpu._lib.Context()
    # put the line at level 1 (coordinates multiples of 2) by giving every level-1 voxel one child
    c = torch.tensor([[0, 0, 0, 0], [0, 2, 0, 0], [0, 4, 0, 0]], dtype=torch.int32).cuda()
    ctx.coords_set(c, 1)
    assert _np(ctx.level_coords(1)).tolist() == [[0, 0, 0, 0], [0, 2, 0, 0], [0, 4, 0, 0]]
    f = torch.zeros((3, 32))
    f[:, 0] = torch.tensor([1.0, 10.0, 100.0])
    k = torch.zeros((27, 32, 32))
    k[:, 0, 0] = torch.arange(27, dtype=torch.float32)
    out = _np(ctx.conv(1, 1, 3, f, k))[:, 0]
    assert np.allclose(out, [1 * 13 + 10 * 14, 1 * 12 + 10 * 13 + 100 * 14, 10 * 12 + 100 * 13])


# ------------------------------------------------------------------------------------ full forward (a4-a10)
@pytest.mark.parametrize("name", H.CASES)
def test_forward_matches_reference_graph(gpu, name):
    case = H.load_case(name)
    m, _ = _model(gpu, case)
    c4 = case["coords"]
    rng = np.random.default_rng(2)
    order = rng.permutation(len(c4))                           # arbitrary caller row order
    batch = {"coords": torch.from_numpy(c4[order]), "features": torch.ones((len(c4), 1))}
    with torch.no_grad():
        y = m(batch)
    ctx = m.context()
    for lvl, ch in ((3, 64), (7, 128)):
        got = _np(ctx.forward_level_features(lvl, ch))
        perm = H.join_perm(_np(ctx.level_coords(lvl)), case[f"level{lvl}_coords"])
        np.testing.assert_allclose(got[perm], case[f"level{lvl}_feats"], rtol=2e-3, atol=2e-4)
    g = _np(y["global"])
    assert g.shape == case["global"].shape
    assert H.cosine_err(g, case["global"]).max() < 1e-4        # north_star bar
    np.testing.assert_allclose(g, case["global"], rtol=1e-3, atol=1e-4)
    kc = m.keypoint_coords()
    assert len(y["descriptors"]) == len(y["keypoints"]) == len(y["sigma"]) == int(case["n_scans"])
    for b in range(int(case["n_scans"])):
        perm = H.join_perm(_np(kc[b]), case[f"kp_coords_{b}"])
        assert H.cosine_err(_np(y["descriptors"][b])[perm], case[f"descriptors_{b}"]).max() < 1e-4
        np.testing.assert_allclose(_np(y["keypoints"][b])[perm], case[f"keypoints_{b}"], rtol=1e-4, atol=2e-3)
        np.testing.assert_allclose(_np(y["sigma"][b])[perm], case[f"sigma_{b}"], rtol=1e-3, atol=1e-4)
        assert y["sigma"][b].shape[1] == 1 and y["keypoints"][b].shape[1] == 3
    # head switches (reference models/minkgl.py:267-268)
    with torch.no_grad():
        yg = m(batch, disable_local_head=True)
        yl = m(batch, disable_global_head=True)
    assert set(yg.keys()) == {"global"} and set(yl.keys()) == {"descriptors", "keypoints", "sigma"}
    assert torch.equal(yg["global"], y["global"])              # deterministic


@pytest.mark.parametrize("name", ["egonn_cart01_b1", "egonn_cart01_b2", "egonn_cart03_b1", "egonn_cart01_50k_b2"])
def test_compute_embedding_matches_reference_selection(gpu, name):
    """eval/evaluate.py:327-361: quantise -> forward -> 128 lowest-sigma keypoints, ascending."""
    case = H.load_case(name)
    m, _ = _model(gpu, case)
    ex = gpu.DescriptorExtractor(m, n_k=128)
    scans = [torch.from_numpy(case[f"points_{b}"]) for b in range(int(case["n_scans"]))]
    out = ex.extract(scans)
    assert H.cosine_err(_np(out["global"]), case["global"]).max() < 1e-4
    kc = m.keypoint_coords()
    off = m.context().level_batch_offsets(3)
    for b in range(len(scans)):
        n = int(out["count"][b])
        want_c = case[f"topk_coords_{b}"]
        assert n == len(want_c)
        rows = _np(out["rows"][b, :n]).astype(np.int64) - off[b]
        got_c = _np(kc[b])[rows]
        # the ordered list of selected super-voxels; fp32 noise may swap near-equal sigmas, so compare
        # exactly where the reference sigmas are separated by more than the forward tolerance
        ws = case[f"topk_sigma_{b}"]
        gap_ok = np.r_[True, np.diff(ws) > 2e-4] & np.r_[np.diff(ws) > 2e-4, True]
        assert (np.all(got_c == want_c, axis=1) | ~gap_ok).all()      # every swap is explained by a near-tie
        # selected descriptors / keypoints equal the full outputs at those rows
        sel = {tuple(c): i for i, c in enumerate(case[f"kp_coords_{b}"].tolist())}
        ridx = np.array([sel[tuple(c)] for c in got_c.tolist()])
        assert H.cosine_err(_np(out["descriptors"][b, :n]), case[f"descriptors_{b}"][ridx]).max() < 1e-4
        np.testing.assert_allclose(_np(out["keypoints"][b, :n]), case[f"keypoints_{b}"][ridx], rtol=1e-4, atol=2e-3)
    # reference-shaped single-scan API
    g, kp, desc = ex.compute_embedding(scans[0])
    assert g.shape == (1, 256) and kp.shape[1] == 3 and desc.shape[1] == 128 and not kp.is_cuda


def test_select_keypoints_exact_vs_oracle(gpu):
    """integer result: on identical sigma values the selection is bit-exact, ties broken by Z-order."""
    from oracle import egonn_ref as ref
    case = H.load_case("egonn_cart01_b2")
    ctx = gpu._lib.Context()
    ctx.coords_set(torch.from_numpy(case["coords"]).cuda(), 2)
    c3 = _np(ctx.level_coords(3))
    off = ctx.level_batch_offsets(3)
    rng = np.random.default_rng(3)
    sigma = rng.uniform(0.1, 2.0, size=(len(c3), 1)).astype(np.float32)
    sigma[rng.integers(0, len(c3), 300)] = 0.5                 # force ties
    kp = rng.standard_normal((len(c3), 3)).astype(np.float32)
    desc = rng.standard_normal((len(c3), 128)).astype(np.float32)
    for n_k in (128, 16, 4096):
        skp, sdesc, rows, cnt = ctx.select_keypoints(torch.from_numpy(sigma).cuda(), torch.from_numpy(kp).cuda(),
                                                     torch.from_numpy(desc).cuda(), n_k)
        for b in range(2):
            s, e = off[b], off[b + 1]
            want = ref.select_keypoints(sigma[s:e], c3[s:e], n_k) + s
            n = int(cnt[b])
            assert n == len(want)
            assert np.array_equal(_np(rows[b, :n]), want)
            assert np.array_equal(_np(skp[b, :n]), kp[want]) and np.array_equal(_np(sdesc[b, :n]), desc[want])
            assert (_np(rows[b, n:]) == -1).all()


# ------------------------------------------------------------------------------------ edge cases
def test_tiny_and_ragged_batches(gpu):
    from oracle import egonn_ref as ref
    m, w = _model(gpu, seed=21)
    oracle = ref.EgoNNOracle(w, ref.CartesianQuantizer(0.1))
    rng = np.random.default_rng(4)
    # one voxel; two distant voxels; a ragged batch with very different sample sizes
    cases = [np.array([[0, 3, -4, 5]], np.int32),
             np.array([[0, 0, 0, 0], [0, 900, -900, 40]], np.int32)]
    from egonn_amd.synth import lidar_scan
    big = ref.CartesianQuantizer(0.1)(lidar_scan(8, 3000))[0]
    small = ref.CartesianQuantizer(0.1)(lidar_scan(9, 40))[0]
    from oracle import me_ops as ops
    cases.append(ops.batched_coordinates([small, big, small[:1]]))
    for c4 in cases:
        f = np.ones((len(c4), 1), np.float32)
        with torch.no_grad():
            y = m({"coords": torch.from_numpy(c4), "features": torch.from_numpy(f)})
        yo = oracle.forward(c4, f)
        assert H.cosine_err(_np(y["global"]), yo["global"]).max() < 1e-4
        kc = m.keypoint_coords()
        for b in range(len(yo["descriptors"])):
            perm = H.join_perm(_np(kc[b]), yo["keypoint_coords"][b])
            assert H.cosine_err(_np(y["descriptors"][b])[perm], yo["descriptors"][b]).max() < 1e-4
            np.testing.assert_allclose(_np(y["sigma"][b])[perm], yo["sigma"][b], rtol=1e-3, atol=1e-4)
            np.testing.assert_allclose(_np(y["keypoints"][b])[perm], yo["keypoints"][b], rtol=1e-4, atol=2e-3)


def test_empty_scan_inside_batch(gpu):
    m, _ = _model(gpu, seed=22)
    ex = gpu.DescriptorExtractor(m, n_k=64)
    from egonn_amd.synth import lidar_scan
    a, b = torch.from_numpy(lidar_scan(1, 2000)), torch.from_numpy(lidar_scan(2, 2500))
    out = ex.extract([a, torch.zeros((0, 3)), b])
    solo_a, solo_b = ex.extract([a]), ex.extract([b])
    assert int(out["count"][1]) == 0
    assert H.cosine_err(_np(out["global"][[0]]), _np(solo_a["global"])).max() < 1e-5
    assert H.cosine_err(_np(out["global"][[2]]), _np(solo_b["global"])).max() < 1e-5


# ------------------------------------------------------------------------------------ full size (BASELINE configs[1])
def test_full_size_properties_batch16(gpu):
    """50k-pt clouds @ 0.1 m, batch 16 (BASELINE.json configs[1]): properties that need no oracle.
       (1) determinism, (2) batch invariance: every sample's outputs equal the single-sample run
       (eval BN, per-sample ECA/GeM), (3) voxel keys strictly increasing and parents consistent,
       (4) a sampled oracle check on one of the 16 scans."""
    from egonn_amd.synth import lidar_scan
    from oracle import egonn_ref as ref
    m, w = _model(gpu, seed=31)
    ex = gpu.DescriptorExtractor(m, n_k=128)
    scans = [torch.from_numpy(lidar_scan(100 + i, 50_000)).cuda() for i in range(16)]
    out1 = ex.extract(scans)
    ctx = m.context()
    counts = [ctx.level_count(l) for l in range(8)]
    assert all(counts[l] > counts[l + 1] > 0 for l in range(7))
    c0 = _np(ctx.level_coords(0))
    k0 = H.rowkey(c0)
    assert len(np.unique(k0)) == len(k0) and (np.diff(c0[:, 0]) >= 0).all()     # unique, batch-contiguous
    c3 = _np(ctx.level_coords(3))
    assert (c3[:, 1:] % 8 == 0).all()
    par = np.unique(H.rowkey(np.c_[c0[:, :1], (c0[:, 1:] >> 3) << 3]))
    assert np.array_equal(par, np.sort(H.rowkey(c3)))                             # level 3 = floor-parents of level 0
    g1, d1, k1 = out1["global"].clone(), out1["descriptors"].clone(), out1["keypoints"].clone()
    out2 = ex.extract(scans)
    assert torch.equal(g1, out2["global"]) and torch.equal(d1, out2["descriptors"]) and torch.equal(k1, out2["keypoints"])
    for b in (0, 7, 15):
        solo = ex.extract([scans[b]])
        assert H.cosine_err(_np(solo["global"]), _np(g1[[b]])).max() < 1e-5
        n = int(solo["count"][0])
        assert n == int(out1["count"][b]) == 128
        assert H.cosine_err(_np(solo["descriptors"][0]), _np(d1[b])).max() < 1e-4
        np.testing.assert_allclose(_np(solo["keypoints"][0]), _np(k1[b]), atol=2e-3)
    # oracle check of 4 of the 16 full-size scans (C/OpenMP restatement: a second or two each; one of them also on the
    # numpy restatement, which is the independent implementation)
    from oracle import egonn_cpu
    co = egonn_cpu.CpuOracle(w, 0.1)
    off3 = ctx.level_batch_offsets(3)
    for b in (3, 0, 9, 15):
        pc = _np(scans[b])
        g_ref, kp_ref, desc_ref, kc_ref, sig_ref, _ = co.compute_embedding(pc, 128)
        if b == 3:
            g_np, kp_np, desc_np, kc_np, sig_np = ref.compute_embedding_with_sigma(ref.EgoNNOracle(w, ref.CartesianQuantizer(0.1)), pc, 128)
            assert H.cosine_err(g_np, g_ref).max() < 1e-5 and np.allclose(sig_np, sig_ref, rtol=1e-3, atol=1e-5)
        assert H.cosine_err(_np(g1[[b]]), g_ref).max() < 1e-4
        rows = _np(out1["rows"][b]).astype(np.int64)
        got_c = c3[rows]
        same = np.all(got_c[:, 1:] == kc_ref, axis=1)
        assert (same | ~_gap_ok(sig_ref)).all(), b                # every swap is a near-tie of the saliencies
        assert same.sum() >= 64
        assert H.cosine_err(_np(d1[b])[same], desc_ref[same]).max() < 1e-4
        np.testing.assert_allclose(_np(k1[b])[same], kp_ref[same], atol=2e-3)


def test_streamed_batches_equal_sequential(gpu):
    """throughput mode: several batches in flight on separate HIP streams / contexts must give exactly the
    results of running them one after the other (bitwise: same kernels, same order inside a batch)."""
    from egonn_amd.synth import lidar_scan
    m, _ = _model(gpu, seed=41)
    ex = gpu.DescriptorExtractor(m, n_k=64)
    batches = []
    for b in range(5):
        scans = [lidar_scan(500 + 10 * b + i, 4000 + 700 * i) for i in range(3)]
        off = [0]
        for s in scans:
            off.append(off[-1] + len(s))
        batches.append((torch.from_numpy(np.concatenate(scans)).cuda(), off))
    seq = []
    for p, o in batches:
        r = ex.extract_packed(p, o)
        seq.append({k: v.clone() for k, v in r.items()})
    torch.cuda.synchronize()
    for n_streams in (2, 3):
        got = [r for r in ex.extract_stream(iter(batches), n_streams=n_streams)]
        torch.cuda.synchronize()
        for a, b in zip(seq, got):
            for k in ("global", "keypoints", "descriptors", "count", "rows"):
                assert torch.equal(a[k], b[k]), (n_streams, k)


@pytest.mark.parametrize("name", H.MINKLOC_CASES)
def test_minkloc_forward_matches_reference_graph(gpu, name):
    """row a12: MinkFPN backbone + GeM through the per-operator C ABI vs the reference graph fixtures."""
    case = H.load_case(name)
    if str(case["model"]) == "MinkLoc3D":
        mp = gpu.ModelParams(model="MinkLoc3D", coordinates="cartesian", quantization_step=0.3)
    else:
        mp = gpu.ModelParams(model="MinkLoc", coordinates="cartesian", quantization_step=0.3, block=str(case["block"]))
    m = gpu.model_factory(mp)
    w = H.seeded_weights(case["weight_seed"], name)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    m = m.to("cuda").eval()
    c4 = case["coords"]
    order = np.random.default_rng(3).permutation(len(c4))
    y = m({"coords": torch.from_numpy(c4[order]), "features": torch.ones((len(c4), 1))})
    g = _np(y["global"])
    assert g.shape == case["global"].shape and set(y.keys()) == {"global"}
    assert H.cosine_err(g, case["global"]).max() < 1e-4
    np.testing.assert_allclose(g, case["global"], rtol=1e-3, atol=1e-4)
    yt = m.train()({"coords": torch.from_numpy(c4), "features": torch.ones((len(c4), 1))})     # train mode: tests/test_gpu_train.py
    assert yt["global"].requires_grad and yt["global"].shape == y["global"].shape


def test_triplet_loss_matches_oracle(gpu):
    """row a13 (forward + dLoss/dE): HIP batch-hard triplet loss vs the oracle and vs torch autograd of the same
    formula on the CPU."""
    from oracle import egonn_ref as ref
    from egonn_amd.loss import BatchHardTripletLossWithMasks
    rng = np.random.default_rng(5)
    for n, d in ((64, 256), (200, 256), (7, 32)):
        e = (rng.standard_normal((n, d)) * 0.3).astype(np.float32)
        lab = rng.integers(0, max(2, n // 6), n)
        pm = (lab[:, None] == lab[None, :]) & ~np.eye(n, dtype=bool)
        nm = lab[:, None] != lab[None, :]
        pm[0] = False
        et = torch.from_numpy(e).cuda().requires_grad_(True)
        loss, stats, (a, p, q) = BatchHardTripletLossWithMasks(0.2)(et, torch.from_numpy(pm), torch.from_numpy(nm))
        want, wstats, (wa, wp, wq) = ref.batch_hard_triplet_loss(e, pm, nm, 0.2)
        assert np.array_equal(_np(a), wa) and np.array_equal(_np(p), wp) and np.array_equal(_np(q), wq)
        assert abs(float(loss.detach()) - want) < 1e-5
        for k, v in wstats.items():
            assert abs(stats[k] - v) <= 1e-4 * max(1.0, abs(v)), k
        loss.backward()
        ec = torch.from_numpy(e).double().requires_grad_(True)
        D = torch.cdist(ec, ec, p=2)
        ta, tp_, tq = (torch.from_numpy(x) for x in (wa, wp, wq))
        li = torch.relu(D[ta, tp_] - torch.minimum(D[ta, tq], D[tp_, tq]) + 0.2)
        (li[li > 0].mean() if (li > 0).any() else li.sum() * 0).backward()
        np.testing.assert_allclose(_np(et.grad), ec.grad.numpy(), rtol=1e-3, atol=1e-6)


@pytest.mark.parametrize("coordinates,step", [("polar", [1.0, 0.3, 0.2]), ("cartesian", 0.3)])
def test_config0_kitti_shaped_scan(gpu, coordinates, step):
    """BASELINE.json configs[0] as restated in SURVEY.md §8(d): one ~120k-point scan after KITTI-style filtering
    (drop zero points, keep z > -1.5), polar (1 deg, 0.3 m, 0.2 m) and Cartesian 0.3 m, seeded weights, through the
    whole path vs the CPU oracle."""
    from egonn_amd.synth import lidar_scan
    from oracle import egonn_ref as ref
    pc = lidar_scan(1, n_points=120_000)
    pc = pc[~np.all(pc == 0, axis=1)]
    pc = pc[pc[:, 2] > -1.5]
    m, w = _model(gpu, seed=51, coordinates=coordinates, step=step)
    ex = gpu.DescriptorExtractor(m, n_k=128)
    g, kp, desc = ex.compute_embedding(torch.from_numpy(pc))
    q = ref.PolarQuantizer(step) if coordinates == "polar" else ref.CartesianQuantizer(step)
    g_ref, kp_ref, desc_ref, kc_ref, sig_ref = ref.compute_embedding_with_sigma(ref.EgoNNOracle(w, q), pc, 128)
    # voxel sets: the polar quantiser takes atan2 in fp64 rounded once to fp32, like the CPU libraries -> identical bins
    c_gpu = _np(m.quantizer(torch.from_numpy(pc))[0])
    c_ref = q(pc)[0]
    assert set(map(tuple, c_gpu.tolist())) == set(map(tuple, np.asarray(c_ref).tolist()))
    assert H.cosine_err(g, g_ref).max() < 1e-4                 # the north-star bar, Cartesian and polar
    assert kp.shape == kp_ref.shape == (128, 3)
    same = np.isclose(kp.numpy(), kp_ref, atol=2e-3 if coordinates == "cartesian" else 2e-2).all(axis=1)
    assert (same | ~_gap_ok(sig_ref)).all()                    # every swap is a near-tie of the saliencies
    assert H.cosine_err(desc.numpy()[same], desc_ref[same]).max() < 1e-4


def test_config4_database_build_mulran_shaped(gpu):
    """BASELINE.json configs[4] on one GPU: MulRan-shaped scans (64 x 1024 returns, ground cut z > -0.9,
    datasets/mulran/mulran_raw.py:15-25) streamed through DatabaseBuilder; rows keep scan order and equal the
    per-scan extraction."""
    from egonn_amd.synth import lidar_scan
    from egonn_amd.distributed import DatabaseBuilder
    m, _ = _model(gpu, seed=61)
    ex = gpu.DescriptorExtractor(m, n_k=128)

    def load(i):
        pc = lidar_scan(7000 + i, n_points=65_536, n_azimuth=1024)
        return torch.from_numpy(pc[pc[:, 2] > -0.9])

    n = 11
    db = DatabaseBuilder(ex, batch_size=4).build(load, n)
    assert db["global"].shape == (n, 256) and db["count"].shape == (n,) and db["range"] == (0, n)
    for i in (0, 5, 10):
        solo = ex.extract([load(i)])
        assert H.cosine_err(_np(solo["global"]), _np(db["global"][[i]])).max() < 1e-5
        assert int(solo["count"][0]) == int(db["count"][i])


def test_streaming_pipeline_equals_extract(gpu, tmp_path):
    """egonn_amd/stream.py (BASELINE configs[4]): raw MulRan-shaped scans from host buffers AND from .bin files through the
    streaming pipeline (pinned staging by reader threads, device filter writing its offsets on the device, hipGraph step,
    S batches in flight, short last batch, a batch that overflows the reservation -> eager fallback) give bitwise the
    descriptors / keypoints of ScanIngest + extract on the same scans, in order."""
    from egonn_amd.synth import lidar_scan
    from egonn_amd.ingest import ScanIngest
    from egonn_amd.distributed import build_database_streaming
    m, _ = _model(gpu, seed=61)
    ex = gpu.DescriptorExtractor(m, n_k=64)
    rng = np.random.default_rng(5)
    raws = []
    for i in range(11):
        pc = lidar_scan(7100 + i, n_points=20000 + 1500 * (i % 4), n_azimuth=1024)
        pc[rng.integers(0, len(pc), 50)] = 0.0                                  # all-zero returns (dropped by the filter)
        raws.append(np.ascontiguousarray(np.concatenate([pc, rng.random((len(pc), 1), dtype=np.float32)], 1)))
    dense = lidar_scan(7300, n_points=26000, n_azimuth=2048)                     # many more voxels than the calibration batch
    big = [np.ascontiguousarray(np.concatenate([dense * s, np.ones((len(dense), 1), np.float32)], 1)) for s in (1.0, 1.5, 2.0, 2.4)]
    ing = ScanIngest("mulran", m.context().device)

    def want_of(batch):
        pts, off = ing(batch)
        out = ex.extract_packed(pts, off, slot=1)
        return {k: out[k].cpu().clone() for k in ("global", "count", "keypoints", "descriptors")}

    se = gpu.StreamingExtractor(ex, batch_size=4, max_points_per_scan=26000, dataset_type="mulran", slots=3, workers=4, keep_local=True)
    se.calibrate(raws[:4], margin=1.15)
    batches = [raws[0:4], raws[4:8], big, raws[8:11]]                            # third batch overflows, last one is short
    for rnd in range(2):
        got = list(se.run(batches))
        assert len(got) == 4
        for b, g in zip(batches, got):
            w = want_of(b)
            for k in w:
                assert torch.equal(g[k], w[k][: len(b)]), (rnd, k)
    assert se.fallbacks == 2                                                     # the dense batch, once per round
    # .bin files through readinto, and the database build on top (one rank: the all-gather is the identity)
    paths = []
    for i, r in enumerate(raws):
        fn = str(tmp_path / f"{i:04d}.bin"); r.tofile(fn); paths.append(fn)
    se2 = gpu.StreamingExtractor(ex, batch_size=4, max_points_per_scan=26000, dataset_type="mulran", slots=2, workers=2, keep_local=False)
    se2.calibrate(paths[:4], margin=1.5)
    db = build_database_streaming(se2, paths)
    assert db["global"].shape == (11, 256) and db["range"] == (0, 11) and db["fallbacks"] == 0
    want = torch.cat([want_of(raws[i:i + 4])["global"] for i in range(0, 11, 4)])
    assert torch.equal(db["global"].cpu(), want)


def test_knn_and_recall_match_oracle():
    """on-device kNN + recall@k (eval/evaluate.py:80-88) vs the numpy restatement: indices bit-exact (ties by index)."""
    from egonn_amd import retrieval, _lib
    from oracle import retrieval_ref as R
    dev = _lib.require_gpu()
    rng = np.random.default_rng(7)
    m, nq, d, k = 5000, 300, 256, 25
    db = rng.standard_normal((m, d)).astype(np.float32)
    db[100] = db[7]                                                 # exact duplicates -> ties
    db[4000] = db[7]
    qs = (db[rng.integers(0, m, nq)] + 0.05 * rng.standard_normal((nq, d))).astype(np.float32)
    qs[0] = db[7]
    idx, dist = retrieval.knn(torch.from_numpy(qs).to(dev), torch.from_numpy(db).to(dev), k)
    ridx, rdist = R.knn(qs, db, k)
    assert idx.cpu().numpy()[0, :3].tolist() == [7, 100, 4000]
    same = idx.cpu().numpy() == ridx
    # fp32 summation order can swap two neighbours whose distances agree to the last ulp
    assert same.mean() > 0.999
    assert np.allclose(dist.cpu().numpy(), rdist, rtol=1e-5, atol=1e-5)
    mpos = rng.uniform(0, 500, (m, 2)).astype(np.float32)
    qpos = (mpos[ridx[:, 0]] + rng.normal(0, 8, (nq, 2))).astype(np.float32)
    out = retrieval.recall_at_k(torch.from_numpy(db), torch.from_numpy(qs), torch.from_numpy(mpos), torch.from_numpy(qpos),
                                radius=[5, 20], k=k)
    want = R.recall(idx.cpu().numpy(), qpos, mpos, [5, 20], k)
    for r in (5, 20):
        assert np.allclose(out['recall'][r], want[r], atol=1e-9), r
    # k larger than the database, and an empty query set
    i2, d2 = retrieval.knn(torch.from_numpy(qs[:4]).to(dev), torch.from_numpy(db[:3]).to(dev), 5)
    assert (i2[:, 3:] == -1).all() and torch.isinf(d2[:, 3:]).all() and (i2[:, :3] >= 0).all()


def _raw_scans(seeds, n):
    from egonn_amd.synth import lidar_scan
    raws = []
    for s in seeds:
        rng = np.random.default_rng(100 + s)
        pc = lidar_scan(s, n_points=n)
        raw = np.concatenate([pc, rng.uniform(0, 255, (len(pc), 1)).astype(np.float32)], axis=1)
        raw[rng.integers(0, len(raw), 200), :3] = 0.0                 # dropped returns are stored as all-zero points
        raw[rng.integers(0, len(raw), 5), 2] = np.nan
        raws.append(np.ascontiguousarray(raw))
    return raws


def test_ingest_filter_matches_loader_semantics():
    """device zero-point / ground-plane filter == PointCloudLoader.__call__ (misc/point_clouds.py:95-111), bit-exact
    and order-preserving, per scan of a batch; then ingest -> extract == host-filtered -> extract."""
    import egonn_amd
    from egonn_amd import _lib
    from egonn_amd.ingest import ScanIngest
    from oracle import ingest_ref as I
    dev = _lib.require_gpu()
    raws = _raw_scans([3, 4, 5], 30000) + [np.zeros((0, 4), dtype=np.float32)] + _raw_scans([6], 1500)
    for ds in ("mulran", "kitti"):
        pts, off = ScanIngest(ds, dev)(raws)
        assert len(off) == len(raws) + 1 and off[0] == 0
        got = pts.cpu().numpy()
        for b, raw in enumerate(raws):
            want = I.preprocess(I.read_pc(raw), ds)
            assert off[b + 1] - off[b] == len(want), (ds, b)
            assert np.array_equal(got[off[b]:off[b + 1]], want, equal_nan=True), (ds, b)
    # files: .bin payloads read straight into the pinned staging buffer
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        paths = []
        for i, raw in enumerate(raws):
            paths.append(os.path.join(d, f"{i:06d}.bin"))
            raw.tofile(paths[-1])
        p2, o2 = ScanIngest("kitti", dev).load(paths)
        assert o2 == off and torch.equal(torch.nan_to_num(p2), torch.nan_to_num(pts))
    # end to end: the descriptors of the ingested batch equal those of the host-filtered clouds
    mp = egonn_amd.ModelParams(model="egonn", coordinates="cartesian", quantization_step=0.3)
    model = egonn_amd.model_factory(mp)
    w = H.seeded_weights(11)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    ex = egonn_amd.DescriptorExtractor(model.to(dev).eval(), n_k=128)
    raws = _raw_scans([7, 8], 30000)
    pts, off = ScanIngest("mulran", dev)(raws)
    a = ex.extract_packed(pts, off)
    b = ex.extract([torch.from_numpy(I.preprocess(I.read_pc(r), "mulran")) for r in raws])
    assert torch.equal(a["global"], b["global"]) and torch.equal(a["keypoints"], b["keypoints"])


@pytest.mark.parametrize("name", ["egonn_cart01_b2", "egonn_polar_b1"])
def test_bf16_feature_maps_config2(name):
    """BASELINE configs[2] arithmetic: feature maps and sparse-conv weights are bf16 in HBM (fp32 accumulate on
    v_mfma_f32_16x16x32_bf16; dense heads, pooling and outputs fp32).  Stated tolerance vs the reference-graph fixture: global descriptor 1-cos <= 1e-5, local descriptors 1-cos <= 2e-4,
    keypoints <= 5 cm, sigma rtol 2e-2, >= 120 of the 128 selected keypoints in common with the fp32 path."""
    import egonn_amd
    from egonn_amd import _lib
    dev = _lib.require_gpu()
    case = H.load_case(name)
    polar = str(case["coordinates"]) == "polar"
    step = [float(v) for v in case["quantization_step"]]
    mp = egonn_amd.ModelParams(model="egonn", coordinates="polar" if polar else "cartesian",
                               quantization_step=step if polar else step[0])
    model = egonn_amd.model_factory(mp)
    w = H.seeded_weights(int(case["weight_seed"]))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    model = model.to(dev).eval()
    coords = torch.from_numpy(case["coords"]).to(dev)
    batch = {"coords": coords, "features": torch.ones((len(coords), 1), device=dev)}
    y32 = model(batch)
    s32 = [t.clone() for t in y32["sigma"]]
    model.precision = "bf16"
    y = model(batch)
    assert H.cosine_err(y["global"].cpu().numpy(), case["global"]).max() <= 1e-5
    kcs = model.keypoint_coords()
    for b in range(int(case["n_scans"])):
        perm = H.join_perm(kcs[b].cpu().numpy(), case[f"kp_coords_{b}"])
        assert H.cosine_err(y["descriptors"][b].cpu().numpy()[perm], case[f"descriptors_{b}"]).max() <= 2e-4
        assert np.allclose(y["keypoints"][b].cpu().numpy()[perm], case[f"keypoints_{b}"], atol=5e-2)
        assert np.allclose(y["sigma"][b].cpu().numpy()[perm], case[f"sigma_{b}"], rtol=2e-2, atol=1e-5)
        k = min(128, len(s32[b]))
        a = set(torch.topk(s32[b].squeeze(1), k, largest=False).indices.tolist())
        c = set(torch.topk(y["sigma"][b].squeeze(1), k, largest=False).indices.tolist())
        assert len(a & c) >= k - 8
    assert not torch.equal(y["global"], y32["global"])             # the flag really switched the arithmetic
    model.precision = "fp16"
    with pytest.raises(ValueError):
        model(batch)


def _fuzz_cloud(rng, kind, n):
    if kind == 0:
        from egonn_amd.synth import lidar_scan
        return lidar_scan(int(rng.integers(0, 10_000)), n_points=n)
    if kind == 1:                                             # dense blob (many duplicates per voxel)
        return (rng.standard_normal((n, 3)) * np.array([3.0, 3.0, 0.5])).astype(np.float32)
    if kind == 2:                                             # a thin wall: 2-D structure, negative coordinates
        p = rng.uniform(-40, 40, (n, 3)).astype(np.float32)
        p[:, 0] = -7.3 + 0.02 * rng.standard_normal(n).astype(np.float32)
        return p
    p = np.zeros((n, 3), np.float32)                          # a line along x + a few far outliers
    p[:, 0] = rng.uniform(-60, 60, n)
    p[: max(1, n // 50)] = rng.uniform(-150, 150, (max(1, n // 50), 3))
    return p


@pytest.mark.parametrize("seed", range(18))
def test_fuzz_single_scan_vs_c_oracle(gpu, seed):
    """randomised clouds / voxel sizes / sizes through compute_embedding vs the independent C/OpenMP restatement:
    level sizes exact, global descriptor 1-cos <= 1e-4, the selected keypoints (where sigma gaps exceed the forward
    tolerance), their positions and descriptors.  Seeds 12-17 use the polar quantiser (the reference's shipped
    configuration, models/egonn.txt:3-5)."""
    from oracle import egonn_cpu
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([150, 900, 4000, 12000, 30000]))
    polar = seed >= 12
    q = float(rng.choice([0.1, 0.2, 0.35, 0.5]))
    if polar:
        q = [float(rng.choice([0.5, 1.0, 2.0])), float(rng.choice([0.2, 0.3, 0.5])), float(rng.choice([0.2, 0.4]))]
    pc = _fuzz_cloud(rng, seed % 4, n)
    w = H.seeded_weights(50 + seed)
    mp = gpu.ModelParams(model="egonn", coordinates="polar" if polar else "cartesian", quantization_step=q)
    m = gpu.model_factory(mp)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    m = m.to("cuda").eval()
    ex = gpu.DescriptorExtractor(m, n_k=128)
    out = ex.extract([torch.from_numpy(pc)])
    g0, kp0, de0, c0, s0, counts = egonn_cpu.CpuOracle(w, q).compute_embedding(pc, 128)
    ctx = m.context()
    assert [ctx.level_count(l) for l in range(8)] == counts.tolist()
    assert H.cosine_err(_np(out["global"]), g0).max() <= 1e-4
    k = int(out["count"][0])
    assert k == len(c0)
    rows = _np(out["rows"][0, :k]).astype(np.int64)
    got_c = _np(m.keypoint_coords()[0])[rows][:, 1:]
    gap_ok = np.r_[True, np.diff(s0) > 2e-4 * np.maximum(1.0, np.abs(s0[1:]))] & \
        np.r_[np.diff(s0) > 2e-4 * np.maximum(1.0, np.abs(s0[:-1])), True] if k > 1 else np.ones(k, bool)
    same = np.all(got_c == c0, axis=1)
    assert (same | ~gap_ok).all()
    if same.any():
        qmax = max(q) if polar else q
        assert np.allclose(_np(out["keypoints"][0, :k])[same], kp0[same], atol=(2e-2 if polar else 2e-3) + 1e-4 * 8 * qmax)
        assert H.cosine_err(_np(out["descriptors"][0, :k])[same], de0[same]).max() <= 1e-4
