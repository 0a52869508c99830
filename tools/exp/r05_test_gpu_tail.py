"""GPU tests of the resident tail kernel (csrc/tail.hip): levels 5-7 of MinkTrunk + MinkHead + descriptor decoder + pooling
(models/minkgl.py:136-153, 46-60, 207-225; layers/pooling.py:29-86) in ONE launch, against the per-layer launches of the same
library (egonn_debug_set_tail(1)) — the two paths differ by summation order only — and against itself (bitwise: reruns,
batch invariance, eager vs graph, scans too large / too small for the staging rounds).  The per-layer launches are the product
path (the resident kernel is slower: DESIGN.md 3.1e); the fixture / oracle tests of test_gpu_parity.py run on them."""
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


def _model(gpu, seed, step=0.1, **kw):
    mp = gpu.ModelParams(model="egonn", coordinates="cartesian", quantization_step=step)
    m = gpu.model_factory(mp, **kw) if kw else gpu.model_factory(mp)
    w = H.seeded_weights(seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    m = m.to("cuda").eval()
    m.coord_bits = 12
    return m


def _batch(seeds, n_points):
    from egonn_amd.synth import lidar_scan
    scans = [lidar_scan(s, n) for s, n in zip(seeds, n_points)]
    off = [0]
    for s in scans:
        off.append(off[-1] + len(s))
    return torch.from_numpy(np.concatenate(scans)).cuda(), off


def _run(gpu, m, pts, off, tail_mode, slot=0, levels=(5, 6, 7)):
    ex = gpu.DescriptorExtractor(m, n_k=128)
    ctx = m.context(slot)
    ctx.set_tail(tail_mode)
    out = ex.extract_packed(pts, off, slot=slot)
    feats = {l: ctx.forward_level_features(l, 128).clone() for l in levels}
    ctx.plan_status()
    res = {k: v.clone() for k, v in out.items()}
    ctx.set_tail(1)                                   # back to the product path
    return res, feats


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


@pytest.mark.parametrize("sizes", [[20000, 15000, 20000, 12000], [50000, 50000], [3000], [9000, 20000, 500, 3]])
def test_tail_matches_per_layer_launches(gpu, sizes):
    """levels 5-7 feature maps and the global descriptor of the resident kernel vs the per-layer launches: <= 3e-6 of the
    largest value of each map (summation order only); everything the local branch produces is bitwise unchanged."""
    m = _model(gpu, 61)
    pts, off = _batch([900 + i for i in range(len(sizes))], sizes)
    ref, fr = _run(gpu, m, pts, off, 1)
    got, fg = _run(gpu, m, pts, off, 0)
    for l in (5, 6, 7):
        assert fg[l].shape == fr[l].shape
        assert _rel(fg[l], fr[l]) <= 3e-6, f"level {l}: {_rel(fg[l], fr[l])}"
    assert _rel(got["global"], ref["global"]) <= 3e-6
    cos = H.cosine_err(got["global"].cpu().numpy(), ref["global"].cpu().numpy())
    assert cos.max() < 1e-9
    for k in ("keypoints", "descriptors", "count", "rows"):
        assert torch.equal(got[k], ref[k]), k


def test_tail_is_deterministic_and_batch_invariant(gpu):
    """bitwise: the same batch twice; a scan alone vs inside a batch of four (other cluster, other neighbours in flight)."""
    m = _model(gpu, 62)
    pts, off = _batch([910, 911, 912, 913], [20000, 15000, 20000, 12000])
    a, fa = _run(gpu, m, pts, off, 0)
    b, fb = _run(gpu, m, pts, off, 0)
    assert torch.equal(a["global"], b["global"])
    for l in (5, 6, 7):
        assert torch.equal(fa[l], fb[l])
    ctx = m.context(0)
    for i in (0, 2, 3):
        p1, o1 = pts[off[i]:off[i + 1]].contiguous(), [0, off[i + 1] - off[i]]
        s, _ = _run(gpu, m, p1, o1, 0)
        assert torch.equal(s["global"][0], a["global"][i]), f"scan {i} alone differs from the scan in the batch"


@pytest.mark.parametrize("pool", ["GeM", "MAC", "SPoC"])
def test_tail_pooling_modes(gpu, pool):
    mp = gpu.ModelParams(model="egonn", coordinates="cartesian", quantization_step=0.1)
    mp.pooling = pool
    m = gpu.model_factory(mp)
    w = H.seeded_weights(63)
    sd = {k: torch.from_numpy(v) for k, v in w.items() if k in m.state_dict()}
    m.load_state_dict(sd, strict=False)
    m = m.to("cuda").eval()
    m.coord_bits = 12
    pts, off = _batch([920, 921], [20000, 9000])
    ref, _ = _run(gpu, m, pts, off, 1)
    got, _ = _run(gpu, m, pts, off, 0)
    assert _rel(got["global"], ref["global"]) <= 3e-6


def test_tail_coarse_and_dense_scans(gpu):
    """scans whose levels 5-7 are far larger (0.02 m voxels: thousands of rows per scan at level 5 -> several windows, row
    pieces, 32-channel rounds) or nearly empty (0.5 m voxels) than the benchmark's: every staging path against the per-layer
    launches."""
    pts, off = _batch([930, 931], [50000, 30000])
    for step, tol in ((0.02, 3e-6), (0.5, 3e-6)):
        m = _model(gpu, 64, step=step)
        m.coord_bits = 14
        ref, fr = _run(gpu, m, pts, off, 1)
        got, fg = _run(gpu, m, pts, off, 0)
        for l in (5, 6, 7):
            assert fg[l].shape == fr[l].shape
            if fr[l].numel():
                assert _rel(fg[l], fr[l]) <= tol, f"step {step} level {l}: {_rel(fg[l], fr[l])} ({fr[l].shape[0]} rows)"
        assert _rel(got["global"], ref["global"]) <= tol


def test_tail_graph_replay_matches_eager_bitwise(gpu):
    """the captured step replays the resident kernel with monotonic flags (nothing is zeroed between replays): replays of
    three different batches, twice, equal the eager results bit for bit."""
    m = _model(gpu, 65)
    ex = gpu.DescriptorExtractor(m, n_k=128)
    batches = [_batch([700, 701, 702, 703], [20000, 15000, 20000, 12000]),
               _batch([710, 711, 712, 713], [9000, 20000, 20000, 20000]),
               _batch([720, 721, 722, 723], [20000, 500, 18000, 3])]
    m.context(1).set_tail(0)                          # the resident kernel (opt-in), eager ...
    eager = [{k: v.clone() for k, v in ex.extract_packed(p, o, slot=1).items()} for p, o in batches]
    caps = ex.calibrate(batches[0][0], batches[0][1], margin=1.5)
    m.context(0).set_tail(0)                          # ... and in the captured step
    gx = ex.graph(batch_size=4, max_points=80000, level_capacity=caps)
    gx.ctx.set_tail(0)
    for rnd in range(3):
        for (p, o), want in zip(batches, eager):
            out = gx.run(p, o)
            gx.status()
            assert torch.equal(out["global"], want["global"]), f"round {rnd}"


def test_tail_concurrent_streams(gpu):
    """four contexts on four streams, interleaved batches (the bench's protocol): every result equals the single-stream
    result bit for bit — clusters of different launches share the chip without exchanging anything."""
    m = _model(gpu, 66)
    ex = gpu.DescriptorExtractor(m, n_k=128)
    batches = [_batch([800 + 4 * i + j for j in range(4)], [20000, 15000, 18000, 12000]) for i in range(8)]
    for s in range(4):
        m.context(s).set_tail(0)                      # the resident kernel on every stream's context
    want = [ex.extract_packed(p, o, slot=0)["global"].clone() for p, o in batches]
    torch.cuda.synchronize()
    for rnd in range(3):
        outs = [r["global"] for r in ex.extract_stream(batches, n_streams=4)]
        torch.cuda.synchronize()
        for i, (g, w) in enumerate(zip(outs, want)):
            assert torch.equal(g, w), f"round {rnd} batch {i}"
    for s in range(4):
        m.context(s).plan_status()
