"""GPU tests of the capturable pipeline (BASELINE.json configs[2]): reserved plans whose level sizes never reach the
host, the whole step (voxelise -> forward -> top-128) captured once into a hipGraph and replayed on other batches, bf16
feature maps at batch 64, and the precision-explicit sparse-convolution operator with its per-group column sums."""
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


def _model(gpu, seed, precision="fp32"):
    mp = gpu.ModelParams(model="egonn", coordinates="cartesian", quantization_step=0.1)
    m = gpu.model_factory(mp)
    w = H.seeded_weights(seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    m = m.to("cuda").eval()
    m.coord_bits = 12
    m.precision = precision
    return m


def _batch(seeds, n_points):
    from egonn_amd.synth import lidar_scan
    scans = [lidar_scan(s, n) for s, n in zip(seeds, n_points)]
    off = [0]
    for s in scans:
        off.append(off[-1] + len(s))
    return torch.from_numpy(np.concatenate(scans)).cuda(), off


KEYS = ("global", "keypoints", "descriptors", "count", "rows")


def test_graph_replay_matches_eager_bitwise(gpu):
    """one captured hipGraph, three different batches (different clouds AND different sizes): every replay equals the
    eager (size-query) run of the same batch bit for bit; the host never learns the level sizes during a replay."""
    m = _model(gpu, 51)
    ex = gpu.DescriptorExtractor(m, n_k=128)
    batches = [_batch([700, 701, 702, 703], [20000, 15000, 20000, 12000]),
               _batch([710, 711, 712, 713], [9000, 20000, 20000, 20000]),
               _batch([720, 721, 722, 723], [20000, 500, 18000, 3])]
    eager = []
    for p, o in batches:
        eager.append({k: v.clone() for k, v in ex.extract_packed(p, o, slot=1).items()})
    caps = ex.calibrate(batches[0][0], batches[0][1], margin=1.5)
    gx = ex.graph(batch_size=4, max_points=80000, level_capacity=caps)
    for rnd in range(2):                       # second round: pure replays
        for (p, o), want in zip(batches, eager):
            out = gx.run(p, o)
            gx.status()
            for k in KEYS:
                assert torch.equal(out[k], want[k]), (rnd, k)
    assert gx.graph is not None
    # the sizes of the LAST replayed batch are available on request
    assert gx.ctx.level_count(0) == m.context(1).level_count(0)


def test_graph_reports_batches_that_do_not_fit(gpu):
    m = _model(gpu, 52)
    ex = gpu.DescriptorExtractor(m, n_k=32)
    small = _batch([800, 801], [3000, 3000])
    big = _batch([810, 811], [30000, 30000])
    caps = ex.calibrate(small[0], small[1], margin=1.2)
    gx = ex.graph(batch_size=2, max_points=60000, level_capacity=caps)
    gx.run(*small)
    gx.status()
    gx.run(*big)                                # more voxels than reserved: flagged on the device, never out of bounds
    with pytest.raises(RuntimeError, match="reserve"):
        gx.status()
    out = gx.run(*small)                        # the context stays usable
    gx.status()
    want = ex.extract_packed(small[0], small[1], slot=1)
    for k in KEYS:
        assert torch.equal(out[k], want[k]), k
    with pytest.raises(ValueError):
        gx.run(*_batch([1, 2, 3], [10, 10, 10]))   # wrong number of scans


@pytest.mark.parametrize("level", [0, 1, 2, 3, 5])
def test_graph_single_level_overflow_is_clipped(gpu, level):
    """A batch that exceeds the reservation of ONE level only (the others have room): every table builder and conv kernel
    clips to the capacity (nbr27 / row-group builder / sconv group counts), the overflow is reported, nothing
    faults, and the next fitting batch in the same context is bitwise the eager result."""
    m = _model(gpu, 57)
    ex = gpu.DescriptorExtractor(m, n_k=32)
    small = _batch([822, 823], [2500, 2500])    # the same scenes, subsampled: fewer rows at every level
    big = _batch([822, 823], [30000, 30000])
    counts = [c - 1024 for c in ex.calibrate(big[0], big[1], margin=1.0)]      # rows of `big` per level
    small_counts = [c - 1024 for c in ex.calibrate(small[0], small[1], margin=1.0)]
    caps = [int(1.5 * c) + 1024 for c in counts]
    caps[level] = max(int(0.6 * counts[level]), small_counts[level] + 8)       # only this level is too small for `big`
    assert small_counts[level] < caps[level] < counts[level]
    gx = ex.graph(batch_size=2, max_points=60000, level_capacity=caps)
    gx.run(*small)
    gx.status()
    for _ in range(2):
        gx.run(*big)
        with pytest.raises(RuntimeError, match="reserve"):
            gx.status()
    out = gx.run(*small)
    gx.status()
    want = ex.extract_packed(small[0], small[1], slot=1)
    for k in KEYS:
        assert torch.equal(out[k], want[k]), (level, k)


def test_graph_out_of_range_points_are_reported_not_fatal(gpu):
    """A return beyond the coordinate range (or NaN) in a reserved, sync-free plan: reported by status(), the step itself
    stays in bounds (the point is clamped into its sample's range, never given a key of a sample beyond the batch), and
    the context keeps working."""
    m = _model(gpu, 58)
    ex = gpu.DescriptorExtractor(m, n_k=32)
    good = _batch([830, 831], [6000, 6000])
    caps = ex.calibrate(good[0], good[1], margin=1.5)
    gx = ex.graph(batch_size=2, max_points=12000, level_capacity=caps)
    gx.run(*good)
    gx.status()
    bad = good[0].clone()
    bad[17] = torch.tensor([250.0, -300.0, 900.0])          # +-204.8 m is the range at 0.1 m voxels / 12 coordinate bits
    bad[7000] = torch.tensor([float("nan"), 0.0, 1.0])
    for _ in range(2):
        gx.run(bad, good[1])
        with pytest.raises(RuntimeError, match="range"):
            gx.status()
    out = gx.run(*good)
    gx.status()
    want = ex.extract_packed(good[0], good[1], slot=1)
    for k in KEYS:
        assert torch.equal(out[k], want[k]), k


def test_config2_bf16_batch64_graph(gpu):
    """BASELINE configs[2]: bf16 feature maps, batch 64 x 50k points, on-device quantisation, captured forward.
    Stated tolerance against the fp32 path on the same clouds: global descriptor 1-cos <= 2e-4, selected local
    descriptors (matched by keypoint row) 1-cos <= 2e-3, at least 85 % of the 128 selected keypoints in common; the
    integer work (voxel counts per level, sample offsets) is identical; replays are bitwise reproducible."""
    B = 64
    p, o = _batch(list(range(2000, 2000 + B)), [50_000] * B)
    m32 = _model(gpu, 61, "fp32")
    ex32 = gpu.DescriptorExtractor(m32, n_k=128)
    ref = {k: v.clone() for k, v in ex32.extract_packed(p, o).items()}
    counts32 = [m32.context().level_count(l) for l in range(8)]
    m16 = _model(gpu, 61, "bf16")
    ex16 = gpu.DescriptorExtractor(m16, n_k=128)
    caps = ex16.calibrate(p, o, margin=1.2)
    gx = ex16.graph(batch_size=B, max_points=B * 50_000, level_capacity=caps)
    out = {k: v.clone() for k, v in gx.run(p, o).items()}
    gx.status()
    assert [gx.ctx.level_count(l) for l in range(8)] == counts32
    out2 = gx.run(p, o)
    gx.status()
    for k in KEYS:
        assert torch.equal(out[k], out2[k]), k
    g16, g32 = out["global"].cpu().numpy(), ref["global"].cpu().numpy()
    assert H.cosine_err(g16, g32).max() <= 2e-4
    assert (out["count"] == 128).all() and (ref["count"] == 128).all()
    common = []
    for b in range(B):
        r16, r32 = out["rows"][b].cpu().numpy(), ref["rows"][b].cpu().numpy()
        both, i16, i32 = np.intersect1d(r16, r32, return_indices=True)
        common.append(len(both))
        d16, d32 = out["descriptors"][b].cpu().numpy()[i16], ref["descriptors"][b].cpu().numpy()[i32]
        assert H.cosine_err(d16, d32).max() <= 2e-3
        assert np.abs(out["keypoints"][b].cpu().numpy()[i16] - ref["keypoints"][b].cpu().numpy()[i32]).max() <= 5e-2
    assert np.mean(common) >= 0.85 * 128, np.mean(common)
    # three of the 64 scans against the CPU restatement (oracle/egonn_cpu.c, fp32) — the stated bf16 tolerance of this
    # configuration: global descriptor 1-cos <= 3e-4; >= 94 % of the 128 selected keypoints are the oracle's (the others
    # are saliency near-ties decided differently by bf16 maps); on those, local descriptors 1-cos <= 3e-3 and keypoint
    # positions within 0.05 m (1/16 of the 0.8 m super-voxel)
    from oracle import egonn_cpu
    co = egonn_cpu.CpuOracle(H.seeded_weights(61), 0.1)
    c3 = gx.ctx.level_coords(3).cpu().numpy()
    pts, offs = p.cpu().numpy(), list(o)
    for b in (0, 29, 63):
        g_ref, kp_ref, desc_ref, kc_ref, sig_ref, cnt_ref = co.compute_embedding(pts[offs[b]:offs[b + 1]], 128)
        assert H.cosine_err(g16[[b]], g_ref).max() <= 3e-4, b
        rows = out["rows"][b].cpu().numpy().astype(np.int64)
        key16 = H.rowkey(np.c_[np.zeros(len(rows), np.int64), c3[rows][:, 1:]])
        keyref = H.rowkey(np.c_[np.zeros(len(kc_ref), np.int64), kc_ref])
        both, i16, iref = np.intersect1d(key16, keyref, return_indices=True)
        assert len(both) >= 0.94 * 128, (b, len(both))                # measured: 128 / 126 / 127
        # the sigma-gap rule with a bf16-sized gap (the fp32 tests use 2e-4): an oracle keypoint may be missing from the bf16
        # selection only if its saliency is within BF16_GAP (relative) of the selection threshold = the oracle's 128th sigma
        BF16_GAP = 5e-3                                               # measured: 0 / 1.0e-3 / 0 on the three scans
        sref = np.asarray(sig_ref, dtype=np.float64).reshape(-1)
        missing = np.setdiff1d(np.arange(len(keyref)), iref)
        slack = (sref[-1] - sref[missing]) / np.maximum(1.0, np.abs(sref[missing])) if len(missing) else np.zeros(1)
        print(f"bf16 batch-64 scan {b}: {len(both)} of 128 keypoints shared, largest saliency margin of a missing one {slack.max():.2e}")
        assert slack.max() <= BF16_GAP, (b, slack.max())
        assert H.cosine_err(out["descriptors"][b].cpu().numpy()[i16], desc_ref[iref]).max() <= 3e-3, b
        assert np.abs(out["keypoints"][b].cpu().numpy()[i16] - kp_ref[iref]).max() <= 5e-2, b


def test_sparse_conv_precisions_and_group_sums(gpu):
    """egonn_sparse_conv: fp32 == plain kernel to rounding; bf16 maps within bf16 rounding of the fp32 result computed on
    the SAME rounded inputs; per-group column sums add up to the per-sample column sums of the stored output."""
    case = H.load_case("egonn_cart01_b2")
    ctx = gpu._lib.Context()
    ctx.coords_set(torch.from_numpy(case["coords"]).cuda(), 2)
    ref = gpu._lib.Context()
    ref.coords_set(torch.from_numpy(case["coords"]).cuda(), 2)
    ref.set_naive_conv(True)
    torch.manual_seed(5)
    for kind, lvl, ci, co in [(0, 1, 32, 32), (0, 2, 32, 64), (0, 3, 64, 64), (0, 4, 128, 128), (1, 2, 32, 32),
                              (1, 5, 128, 128), (2, 3, 64, 64), (2, 5, 128, 128), (0, 5, 256, 256), (2, 4, 128, 256)]:
        lin = lvl if kind == 0 else (lvl - 1 if kind == 1 else lvl + 1)
        K = 27 if kind == 0 else 8
        x = torch.randn(ctx.level_count(lin), ci, device="cuda")
        w = torch.randn(K, ci, co, device="cuda") / np.sqrt(ci * (9 if K == 27 else 2))
        sc, sh = torch.rand(co, device="cuda") + 0.5, torch.randn(co, device="cuda") * 0.1
        want = ref.sparse_conv(kind, lvl, x, w, sc, sh, relu=True)
        got, sums = ctx.sparse_conv(kind, lvl, x, w, sc, sh, relu=True, group_sums=True)
        scale = want.abs().max()
        assert ((got - want).abs().max() / scale) < 2e-5, (kind, lvl, ci, co)
        # group sums: the groups of sample b are first[b]..first[b+1]
        ng, first = ctx.map_groups(kind, lvl)
        off = ctx.level_batch_offsets(lvl)
        for b in range(2):
            s_groups = sums[first[b]:first[b + 1]].double().sum(0)
            s_rows = got[off[b]:off[b + 1]].double().sum(0)
            assert torch.allclose(s_groups, s_rows, rtol=1e-5, atol=1e-3), (kind, lvl, b)
        # bf16 maps: inputs and kernel rounded to bf16, fp32 accumulation, output rounded to bf16
        xb = x.to(torch.bfloat16)
        wb = w.to(torch.bfloat16).float()
        want16 = ref.sparse_conv(kind, lvl, xb.float().contiguous(), wb, sc, sh, relu=True)
        got16 = ctx.sparse_conv(kind, lvl, xb.contiguous(), w, sc, sh, relu=True)
        assert got16.dtype == torch.bfloat16
        err = (got16.float() - want16).abs().max() / want16.abs().max()
        assert err < 6e-3, (kind, lvl, ci, co, float(err))        # one bf16 rounding of the output (2^-8 relative)
        # both MFMA kernels (per-wave / workgroup-cooperative) agree to summation order in fp32, to one bf16 ulp in bf16
        per_wave = None
        for var in (2, 4, 16, 32):
            ctx.lib.egonn_debug_set_naive_conv(ctx.h, var)
            v32, s32 = ctx.sparse_conv(kind, lvl, x, w, sc, sh, relu=True, group_sums=True)
            v16 = ctx.sparse_conv(kind, lvl, xb.contiguous(), w, sc, sh, relu=True)
            assert ((v32 - got).abs().max() / scale) < 2e-5, (var, kind, lvl)
            assert ((v16.float() - got16.float()).abs().max() / scale) < 8e-3, (var, kind, lvl)
            if var == 2:
                per_wave = (v32, s32)
            if var in (16, 32):     # the LDS-DMA kernel (KSP-wave / 4-wave workgroups) runs the register-ring kernel's arithmetic
                assert torch.equal(v32, per_wave[0]) and torch.equal(s32, per_wave[1]), (var, kind, lvl)
        ctx.lib.egonn_debug_set_naive_conv(ctx.h, 0)


def test_ignore_keypoint_saliency_draws_random_keypoints(gpu):
    """eval/evaluate.py:354-356: with ignore_keypoint_saliency the n_k keypoints are a random subset, otherwise the n_k
    lowest sigmas in ascending order (get_keypoints_idxes works on the sigma tensor it is given)."""
    m = _model(gpu, 53)
    sig = torch.rand(1000, 1)
    ex = gpu.DescriptorExtractor(m, n_k=128)
    idx = ex.get_keypoints_idxes(sig, 128)
    want = torch.topk(sig.squeeze(1), 128, largest=False).indices
    assert torch.equal(torch.sort(sig[idx, 0]).values, sig[idx, 0]) and set(idx.tolist()) == set(want.tolist())
    assert len(ex.get_keypoints_idxes(torch.rand(50, 1), 128)) == 50
    exr = gpu.DescriptorExtractor(m, n_k=128, ignore_keypoint_saliency=True)
    r1, r2 = exr.get_keypoints_idxes(sig, 128), exr.get_keypoints_idxes(sig, 128)
    assert len(set(r1.tolist())) == 128 and set(r1.tolist()) != set(want.tolist()) and not torch.equal(r1, r2)
    p, o = _batch([900, 901], [8000, 8000])
    a, b = ex.extract_packed(p, o), exr.extract_packed(p, o, slot=1)
    assert torch.equal(a["global"], b["global"]) and not torch.equal(a["rows"], b["rows"])
    assert (b["count"] == 128).all()


def test_mac_and_spoc_pooling(gpu):
    """PoolingWrapper 'MAC' / 'SPoC' (layers/pooling.py:46-69): per-scan max / mean of the decoded global feature rows;
    everything in front of the pooling is the GeM model's (the GeM exponent is the only parameter that differs)."""
    from egonn_amd import model as M
    w = H.seeded_weights(71)
    outs = {}
    for method in ("GeM", "MAC", "SPoC"):
        mp = gpu.ModelParams(model="egonn", coordinates="cartesian", quantization_step=0.2)
        q = mp.quantizer
        gl = [M.PLANES[i - 1] for i in M.GLOBAL_LEVELS]
        ll = [M.PLANES[i - 1] for i in M.LOCAL_LEVELS]
        m = M.MinkGL(M.MinkTrunk(in_channels=1, planes=M.PLANES, conv0_kernel_size=5),
                     local_head=M.MinkHead(M.LOCAL_LEVELS, ll, M.LOCAL_CH), local_descriptor_size=M.LOCAL_DIM, local_normalize=True,
                     global_head=M.MinkHead(M.GLOBAL_LEVELS, gl, M.GLOBAL_CH), global_descriptor_size=M.GLOBAL_DIM,
                     global_pool_method=method, global_normalize=False, quantizer=q)
        sd = {k: torch.from_numpy(v) for k, v in w.items() if k in m.state_dict()}
        assert ("global_pooling.pooling.p" in m.state_dict()) == (method == "GeM")
        m.load_state_dict(sd)
        m = m.to("cuda").eval()
        p, o = _batch([930, 931, 932], [9000, 7000, 4000])
        ctx = m.context()
        ctx.voxelize(p, o, q.mode, q.step)
        outs[method] = m._forward_on_plan(ctx, None)["global"].cpu().numpy()
        # the numpy restatement of the same graph with the same pooling (oracle/egonn_ref.py: mac / spoc / gem)
        from oracle import egonn_ref as R
        c0 = ctx.level_coords(0).cpu().numpy()
        wd = dict(w)
        want = R.EgoNNOracle(wd, R.CartesianQuantizer(0.2)).forward(c0, np.ones((len(c0), 1), np.float32),
                                                                    disable_local_head=True, pool_method=method)["global"]
        assert want.shape == outs[method].shape
        assert H.cosine_err(outs[method], want).max() < 1e-4, method
        np.testing.assert_allclose(outs[method], want, rtol=2e-3, atol=2e-4 * np.abs(want).max(), err_msg=method)
    g, mx, av = outs["GeM"], outs["MAC"], outs["SPoC"]
    assert np.isfinite(mx).all() and np.isfinite(av).all()
    assert (mx >= av - 1e-6).all()                     # max >= mean, per scan and channel
    # GeM with p = 3 on clamped rows lies between the mean of the clamped rows and their max
    assert (g <= np.maximum(mx, 1e-6) + 1e-5).all() and (g >= np.maximum(av, 0) - 1e-5).all()


def _topk_rows(gpu, sigma, offsets, k):
    """egonn_topk_rows through the C-ABI: (rows [B,k] int32 with -1 padding, counts [B])"""
    from egonn_amd import _lib
    lib = _lib.load()
    B = len(offsets) - 1
    sg = torch.from_numpy(sigma).cuda()
    boff = torch.tensor(offsets, dtype=torch.int32, device="cuda")
    rows = torch.empty((B, k), dtype=torch.int32, device="cuda")
    cnt = torch.empty((B,), dtype=torch.int32, device="cuda")
    _lib.check(lib.egonn_topk_rows(sg.data_ptr(), boff.data_ptr(), B, k, rows.data_ptr(), cnt.data_ptr(), _lib._stream()))
    torch.cuda.synchronize()
    return rows.cpu().numpy(), cnt.cpu().numpy()


@pytest.mark.parametrize("k", [1, 128, 500, 513, 2048])
def test_topk_radix_select_edge_cases(gpu, k):
    """The radix-select top-k against a stable argsort (eval/evaluate.py:359 torch.topk(largest=False); ties by row):
    ragged scans incl. an empty one and one shorter than k, a scan longer than the 8192 keys a workgroup caches, heavy
    ties (a handful of distinct values, all-equal, +-0, negatives, infinities) and both ordering paths (k <= 512 counted,
    k > 512 bitonic)."""
    rng = np.random.default_rng(100 + k)
    sizes = [0, 5, 127, 1000, 8192, 8193, 23001, 300]
    parts = []
    for i, n in enumerate(sizes):
        kind = i % 4
        if kind == 0:
            v = rng.uniform(0.01, 2.0, n)
        elif kind == 1:
            v = rng.choice(np.array([0.25, 0.5, 0.5000001, 1.0, -0.0, 0.0, -3.0, np.inf, -np.inf]), n)   # heavy ties
        elif kind == 2:
            v = np.full(n, 0.75)                                                                       # one value: order = rows
        else:
            v = np.round(rng.standard_normal(n), 2)                                                     # ~600 distinct values
        parts.append(v.astype(np.float32))
    sigma = np.concatenate(parts) if parts else np.zeros(0, np.float32)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    rows, cnt = _topk_rows(gpu, sigma, off.tolist(), k)
    for b, n in enumerate(sizes):
        seg = sigma[off[b]:off[b + 1]]
        # the library's order: monotone bits of the float (so -0.0 sorts before +0.0), then the row
        u = seg.view(np.uint32).astype(np.uint64)
        bits = np.where(u & 0x80000000, ~u & 0xFFFFFFFF, u | 0x80000000)
        want = np.argsort((bits << np.uint64(32)) | np.arange(n, dtype=np.uint64), kind="stable")[:k]
        kk = min(k, n)
        assert cnt[b] == kk
        assert np.array_equal(rows[b, :kk], want + off[b]), (b, n)
        assert np.all(rows[b, kk:] == -1)
        # and the values agree with torch.topk (which leaves the order of equal values open)
        if kk:
            tv = torch.topk(torch.from_numpy(seg), kk, largest=False).values.numpy()
            assert np.array_equal(seg[rows[b, :kk] - off[b]], tv)


@pytest.mark.gpu
def test_split_conv_matches_exact_fp32(gpu):
    """The split kernels (sconv_split.hip: fp32 operands as hi + lo fp16, weights scaled by a power of two per kernel, three
    products on the fp16 matrix pipe, fp32 accumulate) against the exact fp32 kernels on every instantiated channel plan and
    map kind, with uneven channel scales (activations from 1e-3 to 1e+2): the maximum deviation from the plain
    one-thread-per-output kernel, relative to the largest output, stays below 3e-6 (the exact MFMA kernel itself: 3e-7 ..
    1.2e-6 — summation-order noise of fp32 accumulation); the decompositions of the split kernel (workgroups of 4 / 8 waves,
    one / two column parts) are bitwise identical, reruns are bitwise identical, the per-group column sums match; a
    kernel with huge or tiny weights (max |W| = 3e4 / 3e-6) keeps the bound (the pack scale is per kernel)."""
    B = 4
    from egonn_amd.synth import lidar_scan
    scans = [lidar_scan(300 + i, 30000) for i in range(B)]
    off = [0]
    for s in scans:
        off.append(off[-1] + len(s))
    pts = torch.from_numpy(np.concatenate(scans)).cuda()
    ctx = gpu._lib.Context(coord_bits=12)
    ctx.voxelize(pts, off, 0, [0.1])
    ref = gpu._lib.Context(coord_bits=12)
    ref.voxelize(pts, off, 0, [0.1])
    ref.set_naive_conv(True)
    for mc in (0, 1):                  # (the offset parts of the small maps have their own test: tests/test_gpu_ksplit.py)
        for lv in range(8):
            ctx.set_ksplit(mc, lv, kparts=1, kw=0, col_parts=0)
    torch.manual_seed(11)
    worst = 0.0
    plans = [(0, 1, 32, 32), (1, 1, 32, 32), (0, 2, 32, 64), (0, 2, 64, 64), (1, 3, 64, 64), (2, 3, 64, 64), (0, 3, 64, 128),
             (0, 4, 128, 128), (1, 5, 128, 128), (2, 5, 128, 128), (0, 3, 64, 32), (0, 4, 128, 64)]
    for kind, lvl, ci, co in plans:
        lin = lvl if kind == 0 else (lvl - 1 if kind == 1 else lvl + 1)
        K = 27 if kind == 0 else 8
        x = torch.randn(ctx.level_count(lin), ci, device="cuda") * torch.exp(torch.randn(ci, device="cuda"))   # uneven channel scales
        w = torch.randn(K, ci, co, device="cuda") / np.sqrt(ci * (9 if K == 27 else 2))
        sc, sh = torch.rand(co, device="cuda") + 0.5, torch.randn(co, device="cuda") * 0.1
        want = ref.sparse_conv(kind, lvl, x, w, sc, sh, relu=False)
        scale = float(want.abs().max())
        ctx.lib.egonn_debug_set_naive_conv(ctx.h, 16)              # exact fp32 MFMA kernel
        exact, exact_sums = ctx.sparse_conv(kind, lvl, x, w, sc, sh, relu=False, group_sums=True)
        ctx.lib.egonn_debug_set_naive_conv(ctx.h, 1142)            # split, lock-step workgroups of 4 waves (the product kernel)
        got, sums = ctx.sparse_conv(kind, lvl, x, w, sc, sh, relu=False, group_sums=True)
        again, _ = ctx.sparse_conv(kind, lvl, x, w, sc, sh, relu=False, group_sums=True)
        assert torch.equal(got, again), (kind, lvl, ci, co)
        e_split = float((got - want).abs().max()) / scale
        e_exact = float((exact - want).abs().max()) / scale
        worst = max(worst, e_split)
        assert e_split < 3e-6, (kind, lvl, ci, co, e_split, e_exact)
        assert float((got - exact).abs().max()) / scale < 4e-6, (kind, lvl, ci, co)
        assert torch.allclose(sums.double().sum(0), got.double().sum(0), rtol=1e-5, atol=1e-2 * max(scale, 1.0))
        # other decompositions of the same arithmetic: 8-wave workgroups, one / two column parts per task
        for var in (1182, 1542, 1942):
            ctx.lib.egonn_debug_set_naive_conv(ctx.h, var)
            v, s2 = ctx.sparse_conv(kind, lvl, x, w, sc, sh, relu=False, group_sums=True)
            assert torch.equal(v, got) and torch.equal(s2, sums), (var, kind, lvl, ci, co)
    # activations spanning 1e-3 .. 1e+3 in one row, weights far from 1: the pack scale follows max |W|
    ctx.lib.egonn_debug_set_naive_conv(ctx.h, 1142)
    for wmag in (3e4, 1.0, 3e-6):
        x = torch.randn(ctx.level_count(2), 64, device="cuda") * torch.logspace(-3, 3, 64, device="cuda")
        w = torch.randn(27, 64, 64, device="cuda")
        w = w / w.abs().max() * wmag
        want = ref.sparse_conv(0, 2, x, w)
        err = float((ctx.sparse_conv(0, 2, x, w) - want).abs().max()) / float(want.abs().max())
        assert err < 3e-6, (wmag, err)
        worst = max(worst, err)
    ctx.lib.egonn_debug_set_naive_conv(ctx.h, 0)
    print(f"split-fp16 worst relative deviation from the plain fp32 kernel: {worst:.2e}")


def test_database_build_rccl_two_gpus(gpu, tmp_path):
    """BASELINE configs[4] over RCCL: 2 ranks, one GPU each, contiguous scan shards, ONE all-gather of the global descriptors;
    every rank ends with the single-process matrix (bitwise: the same kernels on the same scans).  Needs >= 2 GPUs — skipped
    on the 1-GPU boxes, lights up on the first multi-GPU node."""
    if torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one GPU per rank: >= 2 GPUs")
    import socket
    import torch.multiprocessing as tmp_mp
    from egonn_amd.distributed import DatabaseBuilder
    from egonn_amd.synth import lidar_scan
    n_scans = 7
    m = _model(gpu, 91)
    ex = gpu.DescriptorExtractor(m, n_k=32)
    want = DatabaseBuilder(ex, batch_size=3).build(lambda i: torch.from_numpy(lidar_scan(7000 + i, 5000 + 300 * i)), n_scans)["global"].cpu()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "db")
    ctx = tmp_mp.get_context("spawn")
    procs = [ctx.Process(target=_db_worker, args=(r, 2, port, out, n_scans)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    for r in range(2):
        got = torch.load(f"{out}.{r}")
        assert got["ranks_seen"] == 2
        assert torch.equal(got["global"], want), r
