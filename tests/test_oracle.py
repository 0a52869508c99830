"""CPU tests: the oracle (oracle/me_ops.py + oracle/egonn_ref.py) against
  (1) hand-derivable known-answer cases for the sparse primitives (the reference has no tests
      and MinkowskiEngine is absent, so these are the only pins for the primitive semantics),
  (2) torch CPU for the quantiser arithmetic (torch is the reference's own arithmetic backend),
  (3) the golden vectors produced by the reference's own graph code (tests/golden/)."""
import numpy as np
import pytest
import torch

from oracle import egonn_ref as ref
from oracle import me_ops as ops
from tests import helpers as H


# ------------------------------------------------------------------ (1) known-answer primitives
def test_kernel_offsets_order_and_even_kernel():
    o3 = ops.kernel_offsets(3, 1)
    assert o3.shape == (27, 3)
    assert tuple(o3[0]) == (-1, -1, -1) and tuple(o3[1]) == (0, -1, -1) and tuple(o3[3]) == (-1, 0, -1)
    assert tuple(o3[13]) == (0, 0, 0) and tuple(o3[26]) == (1, 1, 1)
    o2 = ops.kernel_offsets(2, 4)          # even kernel: non-centred, multiples of the input stride
    assert o2.min() == 0 and o2.max() == 4
    assert tuple(o2[1]) == (4, 0, 0) and tuple(o2[2]) == (0, 4, 0) and tuple(o2[4]) == (0, 0, 4)


def test_conv3_three_voxel_line_known_answer():
    # voxels at x=0,1,2 ; feature = [1],[10],[100] ; kernel[i] = i (1x1 channel)
    c = np.array([[0, 0, 0, 0], [0, 1, 0, 0], [0, 2, 0, 0]], dtype=np.int32)
    f = np.array([[1.0], [10.0], [100.0]], dtype=np.float32)
    k = np.arange(27, dtype=np.float32).reshape(27, 1, 1)
    out = ops.conv_forward(f, k, ops.kernel_map(c, c, 3, 1), 3)
    # offsets along x only: index 12 = (-1,0,0), 13 = centre, 14 = (+1,0,0)
    assert np.allclose(out[:, 0], [1 * 13 + 10 * 14, 1 * 12 + 10 * 13 + 100 * 14, 10 * 12 + 100 * 13])


def test_strided_conv_and_transpose_known_answer():
    # negative coordinates use true floor: -1 -> parent -2 (slot 1), 0 and 1 -> parent 0
    c = np.array([[0, -1, 0, 0], [0, 0, 0, 0], [0, 1, 0, 0], [0, 1, 1, 1]], dtype=np.int32)
    parents = ops.stride_coords(c, 2)
    assert H.sort_rows(parents).tolist() == [[0, -2, 0, 0], [0, 0, 0, 0]]
    f = np.array([[1.0], [2.0], [4.0], [8.0]], dtype=np.float32)
    k = (10.0 ** np.arange(8)).astype(np.float32).reshape(8, 1, 1) / 1000.0   # slot i -> 10^(i-3)
    maps = ops.kernel_map(c, parents, 2, 1)
    out = ops.conv_forward(f, k, maps, len(parents))
    by = {tuple(p): o for p, o in zip(parents.tolist(), out[:, 0].tolist())}
    assert np.isclose(by[(0, -2, 0, 0)], 1.0 * k[1, 0, 0])                      # (-1,0,0) = parent+(1,0,0)
    assert np.isclose(by[(0, 0, 0, 0)], 2.0 * k[0, 0, 0] + 4.0 * k[1, 0, 0] + 8.0 * k[7, 0, 0])
    # transposed conv: every fine voxel receives parent feature * kernel[slot]
    g = np.array([[3.0], [5.0]], dtype=np.float32)          # features on `parents` rows
    up = ops.conv_transpose_forward(g, k, maps, len(c))
    pf = {tuple(p): v for p, v in zip(parents.tolist(), g[:, 0].tolist())}
    assert np.isclose(up[0, 0], pf[(0, -2, 0, 0)] * k[1, 0, 0])
    assert np.isclose(up[1, 0], pf[(0, 0, 0, 0)] * k[0, 0, 0])
    assert np.isclose(up[3, 0], pf[(0, 0, 0, 0)] * k[7, 0, 0])


def _dense_case(seed, box=10, n=260, cin=3, b=2, lo=-5):
    """random sparse set in a box (negative coordinates included), its features, and the same data as a dense volume
    [b, cin, X, Y, Z] whose index 0 is coordinate `lo`"""
    rng = np.random.default_rng(seed)
    c = np.unique(np.c_[rng.integers(0, b, n), rng.integers(lo, lo + box, (n, 3))], axis=0).astype(np.int32)
    f = rng.standard_normal((len(c), cin)).astype(np.float32)
    vol = np.zeros((b, cin, box, box, box), np.float32)
    vol[c[:, 0], :, c[:, 1] - lo, c[:, 2] - lo, c[:, 3] - lo] = f
    return c, f, vol, lo


@pytest.mark.parametrize("k", [3, 5])
def test_conv_matches_dense_conv3d(k):
    """Independent cross-check of oracle/me_ops.conv_forward + kernel_map: the same convolution computed
    by torch.nn.functional.conv3d on the densified volume.  conv3d is a cross-correlation, out[x] = sum_d in[x + d] w[d],
    which is the form A.5 states for ME; with the volume laid out [X, Y, Z] the dense weight is w[co][ci][kx][ky][kz] =
    kernel[kx + k*ky + k*k*kz][ci][co] (first spatial axis fastest).  A transposed / flipped / y-fastest restatement fails
    this test; it does not pin MinkowskiEngine itself (absent), only that the restatement is the convolution it claims."""
    import torch.nn.functional as F
    c, f, vol, lo = _dense_case(100 + k)
    cin, cout = f.shape[1], 4
    rng = np.random.default_rng(7)
    kern = rng.standard_normal((k ** 3, cin, cout)).astype(np.float32)
    out = ops.conv_forward(f, kern, ops.kernel_map(c, c, k, 1), len(c))
    w = torch.from_numpy(kern).reshape(k, k, k, cin, cout).permute(4, 3, 2, 1, 0).contiguous()    # [kz][ky][kx] -> [co][ci][kx][ky][kz]
    dense = F.conv3d(torch.from_numpy(vol).double(), w.double(), padding=k // 2).numpy()
    want = dense[c[:, 0], :, c[:, 1] - lo, c[:, 2] - lo, c[:, 3] - lo]
    np.testing.assert_allclose(out, want, rtol=1e-4, atol=1e-4)
    # an asymmetric kernel makes the check sensitive to the axis order: the y-fastest reading must NOT match
    w_bad = torch.from_numpy(kern).reshape(k, k, k, cin, cout).permute(4, 3, 1, 2, 0).contiguous()
    bad = F.conv3d(torch.from_numpy(vol).double(), w_bad.double(), padding=k // 2).numpy()
    assert np.abs(bad[c[:, 0], :, c[:, 1] - lo, c[:, 2] - lo, c[:, 3] - lo] - out).max() > 0.1


def test_strided_and_transposed_conv_match_dense():
    """k=2,s=2 (non-centred even kernel, floor parents — also for negative coordinates) against conv3d(stride=2) on the volume
    aligned to even coordinates, and the transposed convolution onto the cached fine map against conv_transpose3d(stride=2)
    read back at the fine voxels (A.5 / A.6)."""
    import torch.nn.functional as F
    c, f, vol, lo = _dense_case(31, box=10, lo=-6)         # lo even: dense index 2X' <-> coordinate lo + 2X'
    cin, cout = f.shape[1], 5
    rng = np.random.default_rng(9)
    kern = rng.standard_normal((8, cin, cout)).astype(np.float32)
    parents = ops.stride_coords(c, 2)
    maps = ops.kernel_map(c, parents, 2, 1)
    out = ops.conv_forward(f, kern, maps, len(parents))
    w = torch.from_numpy(kern).reshape(2, 2, 2, cin, cout).permute(4, 3, 2, 1, 0).contiguous()
    dense = F.conv3d(torch.from_numpy(vol).double(), w.double(), stride=2).numpy()
    pi = (parents[:, 1:] - lo) // 2
    np.testing.assert_allclose(out, dense[parents[:, 0], :, pi[:, 0], pi[:, 1], pi[:, 2]], rtol=1e-4, atol=1e-4)
    # transposed: features on the parents, kernel (8, cout, cin2)
    g = rng.standard_normal((len(parents), cout)).astype(np.float32)
    kt = rng.standard_normal((8, cout, 2)).astype(np.float32)
    up = ops.conv_transpose_forward(g, kt, maps, len(c))
    pv = np.zeros((vol.shape[0], cout, 5, 5, 5), np.float32)
    pv[parents[:, 0], :, pi[:, 0], pi[:, 1], pi[:, 2]] = g
    wt = torch.from_numpy(kt).reshape(2, 2, 2, cout, 2).permute(3, 4, 2, 1, 0).contiguous()       # conv_transpose3d: [in][out][kx][ky][kz]
    dense_up = F.conv_transpose3d(torch.from_numpy(pv).double(), wt.double(), stride=2).numpy()
    np.testing.assert_allclose(up, dense_up[c[:, 0], :, c[:, 1] - lo, c[:, 2] - lo, c[:, 3] - lo], rtol=1e-4, atol=1e-4)


def test_sparse_quantize_first_occurrence_and_negative_floor():
    pc = np.array([[0.05, 0.0, 0.0], [-0.05, 0.0, 0.0], [0.06, 0.01, 0.0], [0.31, 0.0, -0.11]], np.float32)
    d, idx = ops.sparse_quantize(pc, 0.1)
    assert d.tolist() == [[0, 0, 0], [-1, 0, 0], [3, 0, -2]]
    assert idx.tolist() == [0, 1, 3]


# ------------------------------------------------------------------ (2) quantiser arithmetic vs torch
def test_cartesian_division_is_true_fp32_division():
    rng = np.random.default_rng(0)
    pc = (rng.uniform(-80, 80, size=(200000, 3))).astype(np.float32)
    for q in (0.1, 0.3, 0.01):
        want = torch.floor(torch.from_numpy(pc) / q).int().numpy()
        got = np.floor(pc / np.float32(q)).astype(np.int32)
        assert np.array_equal(want, got)


def test_polar_conversion_matches_torch():
    rng = np.random.default_rng(1)
    pc = rng.uniform(-80, 80, size=(100000, 3)).astype(np.float32)
    t = torch.from_numpy(pc)
    theta = 180. + torch.atan2(t[:, 1], t[:, 0]) * 180. / np.pi      # reference quantization.py:35
    dist = torch.sqrt(t[:, 0] ** 2 + t[:, 1] ** 2)
    want = (torch.stack([theta, dist, t[:, 2]], dim=1) / torch.tensor([1., 0.3, 0.2])).numpy()
    got = ref.PolarQuantizer([1., 0.3, 0.2]).to_polar(pc)
    # libm differences between torch and numpy are allowed only as <= 1 ulp noise, and must not
    # move any point across a bin edge in this sample
    assert np.allclose(want, got, rtol=3e-7, atol=3.1e-5)   # 2 ulp at theta ~ 256
    frac_moved = (np.floor(want) != np.floor(got)).any(axis=1).mean()
    assert frac_moved < 1e-4


# ------------------------------------------------------------------ (3) golden vectors
@pytest.mark.parametrize("name", H.CASES)
def test_quantizer_matches_reference(name):
    case = H.load_case(name)
    q = H.make_quantizer(case, ref)
    for b in range(int(case["n_scans"])):
        coords, idx = q(case[f"points_{b}"])
        if str(case["coordinates"]) == "cartesian":
            assert np.array_equal(coords, case[f"quant_coords_{b}"])
            assert np.array_equal(idx, case[f"quant_index_{b}"])
        else:   # transcendental: sets may differ by bin-edge points only
            a = set(map(tuple, coords.tolist()))
            g = set(map(tuple, case[f"quant_coords_{b}"].tolist()))
            assert len(a ^ g) <= max(2, len(g) // 1000)


@pytest.mark.parametrize("name", H.CASES)
def test_forward_matches_reference_graph(name):
    case = H.load_case(name)
    oracle = ref.EgoNNOracle(H.seeded_weights(case["weight_seed"]), H.make_quantizer(case, ref))
    c4 = case["coords"]
    y = oracle.forward(c4, np.ones((len(c4), 1), np.float32), return_internals=True)
    lv = y["_levels"]
    for lvl in range(3, 8):
        assert np.array_equal(H.sort_rows(lv.coords[lvl]), case[f"level{lvl}_coords"])
    for lvl in (3, 7):
        perm = H.join_perm(lv.coords[lvl], case[f"level{lvl}_coords"])
        np.testing.assert_allclose(y["_trunk"][lvl][perm], case[f"level{lvl}_feats"], rtol=2e-4, atol=2e-5)
    assert H.cosine_err(y["global"], case["global"]).max() < 1e-6
    np.testing.assert_allclose(y["global"], case["global"], rtol=1e-4, atol=1e-5)
    for b in range(int(case["n_scans"])):
        perm = H.join_perm(y["keypoint_coords"][b], case[f"kp_coords_{b}"])
        np.testing.assert_allclose(y["keypoints"][b][perm], case[f"keypoints_{b}"], rtol=1e-5, atol=2e-4)
        np.testing.assert_allclose(y["sigma"][b][perm], case[f"sigma_{b}"], rtol=1e-4, atol=1e-5)
        assert H.cosine_err(y["descriptors"][b][perm], case[f"descriptors_{b}"]).max() < 1e-6
        # selection: same ordered list of super-voxel coordinates as torch.topk on the reference sigma
        idx = ref.select_keypoints(case[f"sigma_{b}"], case[f"kp_coords_{b}"], 128)
        sel_sigma = case[f"sigma_{b}"].reshape(-1)[idx]
        assert np.array_equal(sel_sigma, case[f"topk_sigma_{b}"])
        if len(np.unique(sel_sigma)) == len(sel_sigma):      # no ties -> order is fully determined
            assert np.array_equal(case[f"kp_coords_{b}"][idx], case[f"topk_coords_{b}"])


@pytest.mark.parametrize("name", H.MINKLOC_CASES)
def test_minkloc_oracle_matches_reference_graph(name):
    """MinkFPN + GeM (models/minkfpn.py, models/minkloc.py, third_party/minkloc3d/minkloc.py)."""
    case = H.load_case(name)
    oracle = ref.MinkLocOracle(H.seeded_weights(case["weight_seed"], name))
    c4 = case["coords"]
    y = oracle.forward(c4, np.ones((len(c4), 1), np.float32))
    assert H.cosine_err(y["global"], case["global"]).max() < 1e-6
    np.testing.assert_allclose(y["global"], case["global"], rtol=1e-4, atol=1e-5)
    perm = H.join_perm(y["_coords"], case["backbone_coords"])
    np.testing.assert_allclose(y["_feats"][perm], case["backbone_feats"].astype(np.float32), rtol=2e-3, atol=2e-3)


def test_triplet_miner_matches_reference_formulas():
    """The in-tree miner of models/loss.py:114-143 is plain torch (its file cannot be imported because of the
    pytorch_metric_learning import at the top), so its formulas are evaluated with torch here and compared with
    the oracle's index choice."""
    rng = np.random.default_rng(0)
    n = 40
    e = rng.standard_normal((n, 16)).astype(np.float32)
    lab = rng.integers(0, 6, n)
    pm = (lab[:, None] == lab[None, :]) & ~np.eye(n, dtype=bool)
    nm = (np.abs(lab[:, None] - lab[None, :]) >= 2)
    pm[3] = False                                              # anchor without positives is dropped
    D = torch.cdist(torch.from_numpy(e), torch.from_numpy(e), p=2)
    tp, tn = torch.from_numpy(pm), torch.from_numpy(nm)
    mm = D.clone(); mm[~tp] = 0
    (hpd, hpi) = torch.max(mm, dim=1)
    mm = D.clone(); mm[~tn] = float('inf')
    (hnd, hni) = torch.min(mm, dim=1)
    keep = torch.any(tp, dim=1) & torch.any(tn, dim=1)
    loss, stats, (a, p, q) = ref.batch_hard_triplet_loss(e, pm, nm, 0.2)
    assert np.array_equal(a, torch.where(keep)[0].numpy())
    assert np.array_equal(p, hpi[keep].numpy()) and np.array_equal(q, hni[keep].numpy())
    assert abs(stats["max_pos_pair_dist"] - hpd.max().item()) < 1e-5
    assert abs(stats["mean_neg_pair_dist"] - hnd.mean().item()) < 1e-5
    # swap + AvgNonZero (Appendix A.9), evaluated with torch
    d_ap, d_an, d_pn = D[a, p], D[a, q], D[p, q]
    li = torch.relu(d_ap - torch.minimum(d_an, d_pn) + 0.2)
    want = li[li > 0].mean().item() if (li > 0).any() else 0.0
    assert abs(loss - want) < 1e-5


def test_retrieval_oracle_known_answer():
    from oracle import retrieval_ref as R
    m = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 2.0], [1.0, 0.0]], dtype=np.float32)      # rows 1 and 3 tie
    q = np.array([[0.9, 0.0], [0.0, 1.9]], dtype=np.float32)
    idx, dist = R.knn(q, m, 3)
    assert idx.tolist() == [[1, 3, 0], [2, 0, 1]]
    assert np.allclose(dist[0], [0.1, 0.1, 0.9], atol=1e-6)
    mpos = np.array([[0, 0], [10, 0], [0, 10], [50, 50]], dtype=np.float32)
    qpos = np.array([[48, 50], [0, 9]], dtype=np.float32)
    rec = R.recall(idx, qpos, mpos, [5, 20], 3)
    assert rec[5] == [0.5, 1.0, 1.0] and rec[20] == [0.5, 1.0, 1.0]
    idx2, _ = R.knn(q, m[:2], 3)                                   # k larger than the database
    assert idx2.tolist() == [[1, 0, -1], [0, 1, -1]]


def test_ingest_oracle_known_answer():
    from oracle import ingest_ref as I
    raw = np.array([[0, 0, 0, 5], [1, 2, -0.9, 1], [1, 2, -0.8999, 1], [0, 0, 1e-9, 0], [3, 4, np.nan, 0], [5, 6, 7, 8]],
                   dtype=np.float32)
    pc = I.preprocess(I.read_pc(raw), "mulran")
    assert pc.tolist() == [[1.0, 2.0, np.float32(-0.8999)], [5.0, 6.0, 7.0]]


def test_c_openmp_oracle_matches_numpy_oracle_and_fixture():
    """oracle/egonn_cpu.c (the CPU baseline bench.py times) vs the numpy oracle and the reference-graph fixture."""
    from oracle import egonn_cpu, egonn_ref as ref
    case = H.load_case("egonn_cart01_b1")
    w = H.seeded_weights(int(case["weight_seed"]))
    pc = case["points_0"]
    co = egonn_cpu.CpuOracle(w, 0.1)
    g, kp, desc, coords, sigma, counts = co.compute_embedding(pc, 128)
    # fixture produced by the reference's own graph code
    assert H.cosine_err(g, case["global"]).max() <= 1e-4
    assert counts[0] == len(case["quant_coords_0"]) and counts[3] == len(case["kp_coords_0"])
    assert np.array_equal(coords, case["topk_coords_0"][:, 1:])                 # the 128 selected super-voxels, in order
    assert np.allclose(sigma, case["topk_sigma_0"], rtol=1e-3, atol=1e-6)
    # numpy oracle: same selection, same keypoints / descriptors
    orc = ref.EgoNNOracle(w, ref.CartesianQuantizer(0.1))
    g2, kp2, desc2, c2 = ref.compute_embedding(orc, pc, 128)
    assert np.array_equal(coords, c2[:, 1:])
    assert np.allclose(kp, kp2, atol=2e-4) and H.cosine_err(desc, desc2).max() <= 1e-5
    assert H.cosine_err(g, g2).max() <= 1e-6


def test_c_oracle_polar_matches_reference_fixture():
    """oracle/egonn_cpu.c with the polar quantiser (the reference's shipped configuration, models/egonn.txt:3-5) against the
    fixture produced by the reference's own graph code, and against the numpy restatement."""
    from oracle import egonn_cpu, egonn_ref as ref
    case = H.load_case("egonn_polar_b1")
    w = H.seeded_weights(int(case["weight_seed"]))
    pc = case["points_0"]
    step = [float(v) for v in case["quantization_step"]]
    g, kp, de, sc, sg, cnt = egonn_cpu.CpuOracle(w, step).compute_embedding(pc, 128)
    assert H.cosine_err(g, case["global"]).max() < 1e-6
    assert cnt[0] == len(case["quant_coords_0"])
    want = {tuple(c): i for i, c in enumerate(case["topk_coords_0"][:, 1:].tolist())}
    gap = np.r_[True, np.diff(case["topk_sigma_0"]) > 1e-5] & np.r_[np.diff(case["topk_sigma_0"]) > 1e-5, True]
    same = np.array([want.get(tuple(c), -1) == i for i, c in enumerate(sc.tolist())])
    assert (same | ~gap).all()
    g2, kp2, de2, kc2, sg2 = ref.compute_embedding_with_sigma(ref.EgoNNOracle(w, ref.PolarQuantizer(step)), pc, 128)
    assert np.array_equal(sc, kc2[:, 1:]) and np.allclose(kp, kp2, atol=1e-4) and np.allclose(sg, sg2, rtol=1e-4, atol=1e-6)
    assert H.cosine_err(de, de2).max() < 1e-6


def test_c_oracle_under_sanitizers(tmp_path):
    """SURVEY.md §5 / VERDICT r2 6-iv: oracle/egonn_cpu.c built with -fsanitize=address,undefined (plain gcc on its one source
    file + the stand-alone driver oracle/egonn_cpu_san_main.c) and run over ragged / tiny / empty / polar inputs: no
    sanitizer report, and the level counts / output sums equal the regular build's."""
    import os, shutil, struct, subprocess
    from oracle import egonn_cpu
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    here = os.path.dirname(egonn_cpu.SRC)
    exe = str(tmp_path / "egonn_cpu_san")
    cmd = ["gcc", "-O1", "-g", "-fopenmp", "-std=c11", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-mavx2", "-mfma",
           "-o", exe, egonn_cpu.SRC, os.path.join(here, "egonn_cpu_san_main.c"), "-lm"]
    subprocess.run(cmd, check=True)
    w = H.seeded_weights(5)
    keys = egonn_cpu.weight_order()
    from egonn_amd.synth import lidar_scan
    rng = np.random.default_rng(3)
    cases = [("lidar4k", lidar_scan(11, 4000), [0.1]), ("lidar_polar", lidar_scan(12, 3000), [1.0, 0.3, 0.2]),
             ("one_point", np.array([[1.0, 2.0, 0.5]], np.float32), [0.1]),
             ("two_far", np.array([[0.0, 0.0, 0.0], [70.0, -60.0, 3.0]], np.float32), [0.1]),
             ("dup_points", np.repeat(rng.standard_normal((5, 3)).astype(np.float32), 40, axis=0), [0.1]),
             ("negatives", (rng.standard_normal((800, 3)) * 3 - 5).astype(np.float32), [0.3])]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               OMP_NUM_THREADS="2")
    for name, pc, step in cases:
        f = tmp_path / (name + ".bin")
        st = (step * 3)[:3]
        with open(f, "wb") as fh:
            fh.write(struct.pack("<qi3fii", len(pc), 1 if len(step) == 3 else 0, *st, 64, len(keys)))
            for k in keys:
                a = np.ascontiguousarray(w[k], dtype=np.float32).reshape(-1)
                fh.write(struct.pack("<q", a.size)); fh.write(a.tobytes())
            fh.write(np.ascontiguousarray(pc, np.float32).tobytes())
        r = subprocess.run([exe, str(f)], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0 and "ERROR" not in r.stderr and "runtime error" not in r.stderr, (name, r.stderr[-2000:])
        g, kp, de, sc, sg, cnt = egonn_cpu.CpuOracle(w, step if len(step) == 3 else step[0]).compute_embedding(pc, 64, 2)
        tok = r.stdout.split()
        assert int(tok[1]) == len(kp) and [int(t) for t in tok[3:11]] == cnt.tolist(), (name, r.stdout)
        assert np.isclose(float(tok[12]), float(g.astype(np.float64).sum()), rtol=1e-4, atol=1e-4), (name, r.stdout)
