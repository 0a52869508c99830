"""GPU tests of the range guard of the fp16-split sparse convolutions (csrc/sconv_split.hip; the reference's arithmetic is fp32:
models/minkgl.py:105 -> ME's fp32 GEMM).  An fp16 operand part holds |x| < 65504: a finite fp32 activation beyond that must not
silently come back as garbage — egonn_plan_status reports EGONN_STATUS_FP16_RANGE, egonn_ctx_set_exact_fp32 selects the exact
kernels, and DescriptorExtractor.compute_embedding falls back on its own.  Small operands (the input-gradient convolutions of a
training step) are scaled by a power of two before they enter the split kernels (egonn_amd/train.py)."""
import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import __graft_entry__ as entry
    entry.build()
    import egonn_amd
    from egonn_amd import _lib
    egonn_amd._lib = _lib
    return egonn_amd


def _plan(gpu, seeds, n_points=20000):
    from egonn_amd.synth import lidar_scan
    scans = [lidar_scan(s, n_points) for s in seeds]
    off = [0]
    for s in scans:
        off.append(off[-1] + len(s))
    pts = torch.from_numpy(np.concatenate(scans)).cuda()
    ctx = gpu._lib.Context(coord_bits=12)
    ctx.voxelize(pts, off, 0, [0.1])
    return ctx, pts, off


@pytest.mark.gpu
@pytest.mark.parametrize("kind,lvl,ci,co", [(0, 2, 64, 64), (0, 1, 32, 32), (0, 4, 128, 128), (1, 3, 64, 64), (0, 5, 128, 128)])
def test_out_of_range_activation_is_reported_and_exact_mode_is_exact(gpu, kind, lvl, ci, co):
    """One channel at 1e5 (and a single element at 7e4, just above the fp16 range): the split path raises the range status, the
    exact mode returns the plain kernel's result; 6e4 (inside the range) passes on the split path within its 3e-6 bound."""
    Lib = gpu._lib
    ctx, pts, off = _plan(gpu, [21, 22])
    ref, _, _ = _plan(gpu, [21, 22])
    ref.set_naive_conv(True)
    lin = lvl if kind == 0 else lvl - 1
    K = 27 if kind == 0 else 8
    g = torch.Generator(device="cuda").manual_seed(5 + lvl)
    w = torch.randn(K, ci, co, device="cuda", generator=g) / np.sqrt(ci * 9)
    base = torch.randn(ctx.level_count(lin), ci, device="cuda", generator=g)
    cases = {"channel at 1e5": base.clone(), "one element at 7e4": base.clone(), "channel at 6e4": base.clone()}
    cases["channel at 1e5"][:, 3] *= 1e5
    cases["one element at 7e4"][base.shape[0] // 2, 5] = 7e4
    cases["channel at 6e4"][:, 3] = torch.sign(base[:, 3]) * 6e4
    for name, x in cases.items():
        want = ref.sparse_conv(kind, lvl, x, w)
        scale = float(want.abs().max())
        ctx.voxelize(pts, off, 0, [0.1])                   # a fresh plan: the flag word is per plan
        got = ctx.sparse_conv(kind, lvl, x, w)
        if name == "channel at 6e4":
            ctx.plan_status()                              # in range: no report
            assert float((got - want).abs().max()) / scale < 3e-6, name
            continue
        with pytest.raises(Lib.Fp16RangeError) as ei:
            ctx.plan_status()
        assert ei.value.code == 6 and not bool(torch.isfinite(got).all()), name
        ctx.voxelize(pts, off, 0, [0.1])
        ctx.set_exact_fp32(True)
        try:
            exact = ctx.sparse_conv(kind, lvl, x, w)
            ctx.plan_status()
        finally:
            ctx.set_exact_fp32(False)
        assert bool(torch.isfinite(exact).all())
        assert float((exact - want).abs().max()) / scale < 3e-6, name


@pytest.mark.gpu
def test_non_finite_input_raises_the_flag_too(gpu):
    ctx, pts, off = _plan(gpu, [23])
    x = torch.randn(ctx.level_count(2), 64, device="cuda")
    x[7, 1] = float("inf")
    w = torch.randn(27, 64, 64, device="cuda") * 0.05
    ctx.sparse_conv(0, 2, x, w)
    with pytest.raises(gpu._lib.Fp16RangeError):
        ctx.plan_status()


@pytest.mark.gpu
def test_small_gradients_keep_their_relative_accuracy(gpu):
    """Input gradients of 1e-7 (below the 2^-14 = 6e-5 where an fp16 part stops being a normal number): dX of the k=3, the strided
    and the transposed convolution through SparseConvFn.backward against the plain kernel on the same operands — relative to the
    largest |dX| the error stays at the split kernels' 3e-6, i.e. ~1e-13 absolute; an unscaled fp16 split would carry 3e-8."""
    from egonn_amd.train import SparseConvFn
    ctx, pts, off = _plan(gpu, [31, 32])
    ref, _, _ = _plan(gpu, [31, 32])
    ref.set_naive_conv(True)
    ctx.prepare_maps(False)
    ref.prepare_maps(False)
    gen = torch.Generator(device="cuda").manual_seed(3)
    for (ks, lin, lout, ci, co, transposed) in [(3, 2, 2, 64, 64, False), (3, 4, 4, 128, 128, False), (2, 2, 3, 64, 64, False),
                                                (2, 4, 3, 64, 64, True)]:
        K = 27 if ks == 3 else 8
        x = torch.randn(ctx.level_count(lin), ci, device="cuda", generator=gen, requires_grad=True)
        kernel = (torch.randn(K, ci, co, device="cuda", generator=gen) / np.sqrt(ci * 9)).requires_grad_(True)
        y = SparseConvFn.apply(x, kernel, ctx, lin, lout, ks, transposed)
        gy = torch.randn(y.shape, device="cuda", generator=gen) * 1e-7 * torch.exp(torch.randn(co, device="cuda", generator=gen))
        (dx,) = torch.autograd.grad(y, x, gy)
        kd = kernel.detach()
        if ks == 3:
            want = ref.conv(lin, lin, 3, gy, kd.flip(0).transpose(1, 2).contiguous())
        elif not transposed:
            want = ref.conv_transpose(lout, gy, kd.transpose(1, 2).contiguous())
        else:
            want = ref.conv(lout, lin, 2, gy, kd.transpose(1, 2).contiguous())
        scale = float(want.abs().max())
        assert scale < 1e-4
        err = float((dx - want).abs().max()) / scale
        assert err < 3e-6, (ks, lin, lout, transposed, err)
        # element-wise: the bulk of the gradient entries keep 5 digits
        big = want.abs() > 1e-3 * scale
        rel = ((dx - want).abs() / want.abs())[big]
        assert float(rel.median()) < 1e-6 and float(rel.max()) < 1e-2, (ks, float(rel.median()), float(rel.max()))
    ctx.plan_status()


@pytest.mark.gpu
def test_split_pipe_against_exact_mode_end_to_end(gpu):
    """The whole extraction on the fp16-split pipe (sparse convolutions of levels <= 5 with their offset parts, the three local heads
    of models/minkgl.py:175-225 on split Linear kernels) against egonn_ctx_set_exact_fp32(1) (every kernel exact fp32) on the same
    scans: global descriptor 1 - cos < 1e-6, every local descriptor 1 - cos < 1e-6, keypoints within 1e-4 m, sigma within 1e-4
    relative — the two differ by fp32-class rounding only, far inside the 1e-4 cosine bar of the north star."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import helpers as H
    from egonn_amd.synth import lidar_scan
    mp = gpu.ModelParams(model="egonn", coordinates="cartesian", quantization_step=0.1)
    m = gpu.model_factory(mp)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in H.seeded_weights(5).items()})
    m = m.to("cuda").eval()
    ex = gpu.DescriptorExtractor(m, n_k=128)
    scans = [torch.from_numpy(lidar_scan(900 + i, 20000)) for i in range(3)]
    ctx = m.context()

    def run():
        out = ex.extract(scans)
        d, k, s = m._last_local
        ctx.plan_status()
        return out["global"].clone(), d.clone(), k.clone(), s.clone()
    g0, d0, k0, s0 = run()
    ctx.set_exact_fp32(True)
    try:
        g1, d1, k1, s1 = run()
    finally:
        ctx.set_exact_fp32(False)
    cos = lambda a, b: 1.0 - (a * b).sum(1) / (a.norm(dim=1) * b.norm(dim=1))
    assert float(cos(g0, g1).max()) < 1e-6
    assert d0.shape == d1.shape and float(cos(d0, d1).max()) < 1e-6
    assert float((k0 - k1).abs().max()) < 1e-4
    assert torch.allclose(s0, s1, rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
def test_level1_features_on_request(gpu):
    """egonn_forward_level_features(1): by default level 1's block output of fp32 maps is never materialised (its tail runs inside
    level 2's strided convolution) and the call says so; after egonn_debug_keep_level_features(ctx, 1) the map exists, and the
    descriptors are bitwise the same either way."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import helpers as H
    from egonn_amd.synth import lidar_scan
    mp = gpu.ModelParams(model="egonn", coordinates="cartesian", quantization_step=0.1)
    m = gpu.model_factory(mp)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in H.seeded_weights(9).items()})
    m = m.to("cuda").eval()
    ex = gpu.DescriptorExtractor(m, n_k=64)
    scans = [torch.from_numpy(lidar_scan(40 + i, 15000)) for i in range(2)]
    ctx = m.context()
    a = ex.extract(scans)
    with pytest.raises(gpu._lib.EgonnError):
        ctx.forward_level_features(1, 32)
    ctx.keep_level_features(True)
    try:
        b = ex.extract(scans)
        f1 = ctx.forward_level_features(1, 32)
    finally:
        ctx.keep_level_features(False)
    assert f1.shape == (ctx.level_count(1), 32) and bool(torch.isfinite(f1).all()) and float(f1.min()) >= 0.0
    for k in ("global", "descriptors", "keypoints"):
        assert torch.equal(a[k], b[k]), k
