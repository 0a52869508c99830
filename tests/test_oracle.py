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
