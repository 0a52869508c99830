"""GPU tests of the offset parts of the small maps' fp32 sparse convolutions (csrc/sconv_split.hip; models/minkgl.py:144-151 with
planes = [.., 128, 128, 128, 128], layers/eca_block.py:56-73, models/minkgl.py:39): levels 3-5 split the 27 (8) offsets of a
map over KW waves per SIMD inside a workgroup (the product rule), or over separate workgroups with a fixed-order reducer launch
(egonn_debug_set_ksplit).  Every setting is held against the plain one-thread-per-output kernel (<= 3e-6 of the largest output,
the bound of the unsplit kernel), against itself (bitwise reruns) and — because the partition is a function of the layer only
— a scan's rows are bitwise the same alone and inside a batch."""
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


def _plan(gpu, seeds, n_points=30000):
    from egonn_amd.synth import lidar_scan
    scans = [lidar_scan(s, n_points) for s in seeds]
    off = [0]
    for s in scans:
        off.append(off[-1] + len(s))
    pts = torch.from_numpy(np.concatenate(scans)).cuda()
    ctx = gpu._lib.Context(coord_bits=12)
    ctx.voxelize(pts, off, 0, [0.1])
    return ctx


LAYERS = [(0, 3, 64, 64), (1, 4, 64, 64), (0, 4, 64, 128), (0, 4, 128, 128), (1, 5, 128, 128), (0, 5, 128, 128), (2, 5, 128, 128),
          (0, 6, 128, 128), (2, 6, 128, 128), (0, 7, 128, 128), (0, 4, 128, 64)]


def _operands(ctx, kind, lvl, ci, co, seed):
    lin = lvl if kind == 0 else (lvl - 1 if kind == 1 else lvl + 1)
    K = 27 if kind == 0 else 8
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn(ctx.level_count(lin), ci, device="cuda", generator=g) * torch.exp(torch.randn(ci, device="cuda", generator=g))
    w = torch.randn(K, ci, co, device="cuda", generator=g) / np.sqrt(ci * (9 if K == 27 else 2))
    sc = torch.rand(co, device="cuda", generator=g) + 0.5
    sh = torch.randn(co, device="cuda", generator=g) * 0.1
    return x, w, sc, sh


@pytest.mark.gpu
def test_offset_parts_match_plain_kernel(gpu):
    """Default rule, in-workgroup parts KW = 2, 3, 4 and separate-workgroup parts (3 / 9 / 27 of the 27 offsets, 2 / 4 / 8 of the 8
    slots, alone and combined with KW) on every small-map layer shape: each within 3e-6 of the plain kernel, bitwise reruns, and the
    per-group column sums (the ECA pooling of layers/eca_block.py:21-36) equal to the stored rows' sums."""
    ctx = _plan(gpu, [300, 301, 302, 303])
    ref = _plan(gpu, [300, 301, 302, 303])
    ref.set_naive_conv(True)
    ctx.lib.egonn_debug_set_naive_conv(ctx.h, 1142)        # the lock-step split kernel on every level
    worst = 0.0
    for li, (kind, lvl, ci, co) in enumerate(LAYERS):
        x, w, sc, sh = _operands(ctx, kind, lvl, ci, co, 100 + li)
        want = ref.sparse_conv(kind, lvl, x, w, sc, sh, relu=True)
        scale = float(want.abs().max())
        mc = 0 if kind == 0 else 1
        settings = [(-1, -1)] + [(1, 0)] + [(1, kw) for kw in (2, 3, 4)] + \
                   [(kp, 0) for kp in ((3, 9, 27) if kind == 0 else (2, 4, 8))] + [((3 if kind == 0 else 2), 2)]
        outs = []
        for kp, kw in settings:
            if kp >= 0:
                ctx.set_ksplit(mc, lvl, kparts=kp, kw=kw, col_parts=0)
            got, sums = ctx.sparse_conv(kind, lvl, x, w, sc, sh, relu=True, group_sums=True)
            again, sums2 = ctx.sparse_conv(kind, lvl, x, w, sc, sh, relu=True, group_sums=True)
            assert torch.equal(got, again) and torch.equal(sums, sums2), (kind, lvl, ci, co, kp, kw)
            err = float((got - want).abs().max()) / scale
            worst = max(worst, err)
            assert err < 3e-6, (kind, lvl, ci, co, kp, kw, err)
            assert torch.allclose(sums.double().sum(0), got.double().sum(0), rtol=1e-5, atol=1e-2 * max(scale, 1.0))
            outs.append(got)
        # the settings differ by summation order only
        for o in outs[1:]:
            assert float((o - outs[0]).abs().max()) / scale < 4e-6
    print("worst offset-part deviation from the plain kernel:", worst)


@pytest.mark.gpu
def test_offset_parts_are_batch_invariant(gpu):
    """A row's sum is (part 0 + part 1) + part 2 ... over a FIXED partition of the offsets: the rows of a scan come out bitwise
    the same whether the scan is voxelised alone or as one of four (default rule and the separate-workgroup parts)."""
    seeds = [410, 411, 412, 413]
    for which in (0, 2):
        for cfg in ("default", "kparts"):
            batch = _plan(gpu, seeds)                  # (fresh contexts: the rule is per context)
            alone = _plan(gpu, [seeds[which]])
            for (kind, lvl, ci, co) in [(0, 4, 128, 128), (0, 5, 128, 128), (1, 5, 128, 128), (0, 3, 64, 64)]:
                if cfg == "kparts":
                    for c in (batch, alone):
                        c.set_ksplit(0 if kind == 0 else 1, lvl, kparts=(3 if kind == 0 else 2), kw=0, col_parts=0)
                lin = lvl if kind == 0 else lvl - 1
                K = 27 if kind == 0 else 8
                g = torch.Generator(device="cuda").manual_seed(7 * lvl + ci)
                w = torch.randn(K, ci, co, device="cuda", generator=g) / np.sqrt(ci * 9)
                # features as a function of the voxel coordinate, so that both plans see the same input rows
                def feats(c):
                    co_ = c.level_coords(lin).float()[:, 1:]
                    base = torch.sin(co_ @ torch.tensor([[0.013], [0.007], [0.019]], device="cuda") +
                                     torch.arange(ci, device="cuda") * 0.37)
                    return base.contiguous()
                ya = alone.sparse_conv(kind, lvl, feats(alone), w)
                yb = batch.sparse_conv(kind, lvl, feats(batch), w)
                cb = batch.level_coords(lvl)
                rows = (cb[:, 0] == which).nonzero().squeeze(1)
                assert torch.equal(cb[rows][:, 1:], alone.level_coords(lvl)[:, 1:])
                assert torch.equal(yb[rows], ya), (cfg, which, kind, lvl)
