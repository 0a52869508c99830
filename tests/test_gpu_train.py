"""GPU parity tests of the training-mode path (BASELINE configs[3]).

* adjoint identities tie every backward kernel to its (parity-tested) forward:  <conv(x; V), G> == <V, dW(x, G)>  and
  <conv(x; W), G> == <x, dX(G; W)>  — size independent, exact up to fp32 summation order;
* batch-statistics BatchNorm / ECA tail / GeM / Linear Functions against torch autograd on the CPU;
* one whole train-mode step (forward, backward, running statistics) against the fixture the REFERENCE graph produced
  on the stand-in ME ops with torch autograd (tests/golden/make_golden.py train).
"""
import zlib

import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def plan():
    import __graft_entry__ as g
    g.build()
    from egonn_amd import _lib
    from egonn_amd.synth import lidar_scan
    dev = _lib.require_gpu()
    ctx = _lib.Context(dev, coord_bits=12)
    pts = [lidar_scan(50 + i, n_points=9000) for i in range(3)]
    off = [0]
    for p in pts:
        off.append(off[-1] + len(p))
    ctx.voxelize(torch.from_numpy(np.concatenate(pts)).to(dev), off, 0, [0.2])
    return ctx


def rnd(shape, seed, dev, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


def rel(a, b):
    return abs(float(a) - float(b)) / max(abs(float(a)), abs(float(b)), 1e-12)


CONV_CASES = [  # (kernel_size, transposed, level_in, level_out, cin, cout)
    (3, False, 1, 1, 32, 32), (3, False, 2, 2, 32, 64), (3, False, 3, 3, 64, 64), (3, False, 4, 4, 64, 128),
    (3, False, 5, 5, 128, 128), (2, False, 0, 1, 32, 32), (2, False, 3, 4, 128, 128), (2, True, 5, 4, 128, 128),
    (2, True, 4, 3, 64, 64), (1, False, 3, 3, 64, 128), (1, False, 5, 5, 128, 96), (3, False, 2, 2, 16, 24),
]


@pytest.mark.parametrize("ks,tr,lin,lout,cin,cout", CONV_CASES)
def test_conv_backward_adjoint_identities(plan, ks, tr, lin, lout, cin, cout):
    from egonn_amd.train import SparseConvFn
    ctx, dev = plan, plan.device
    n_in, n_out = ctx.level_count(lin), ctx.level_count(lout)
    kshape = (cin, cout) if ks == 1 else (ks ** 3, cin, cout)
    x = rnd((n_in, cin), 1, dev).requires_grad_(True)
    W = rnd(kshape, 2, dev, 0.1).requires_grad_(True)
    G = rnd((n_out, cout), 3, dev)
    V = rnd(kshape, 4, dev, 0.1)
    y = SparseConvFn.apply(x, W, ctx, lin, lout, ks, tr)
    assert y.shape == (n_out, cout)
    (y * G).sum().backward()
    # <conv(x; V), G> == <V, dW>   (the convolution is linear in its kernel)
    yv = SparseConvFn.apply(x.detach(), V, ctx, lin, lout, ks, tr)
    lhs, rhs = (yv.double() * G.double()).sum(), (V.double() * W.grad.double()).sum()
    assert rel(lhs, rhs) < 2e-5, (float(lhs), float(rhs))
    # <conv(x; W), G> == <x, dX>   (and linear in its input)
    lhs, rhs = (y.detach().double() * G.double()).sum(), (x.detach().double() * x.grad.double()).sum()
    assert rel(lhs, rhs) < 2e-5, (float(lhs), float(rhs))
    # a second, independent probe of dX: the gradient itself is linear in G and must vanish for G = 0
    assert torch.isfinite(x.grad).all() and torch.isfinite(W.grad).all()


def test_input_layer_weight_gradient(plan):
    from egonn_amd.train import SparseConvFn
    ctx, dev = plan, plan.device
    n0 = ctx.level_count(0)
    W = rnd((125, 1, 32), 5, dev, 0.1).requires_grad_(True)
    G = rnd((n0, 32), 6, dev)
    y = SparseConvFn.apply(None, W, ctx, 0, 0, 5, False)
    (y * G).sum().backward()
    V = rnd((125, 1, 32), 7, dev, 0.1)
    yv = ctx.conv(0, 0, 5, None, V)
    lhs, rhs = (yv.double() * G.double()).sum(), (V.double() * W.grad.double()).sum()
    assert rel(lhs, rhs) < 2e-5
    # explicit features: same kernel through the general (feature-gathering) path
    ones = torch.ones((n0, 1), device=dev)
    # (unit features run on the bf16 matrix pipe with dout split exactly into three bf16 parts, explicit features on the plain
    #  fp32 kernel: the same sums in another order)
    dk = ctx.conv_backward_weight(0, 0, 5, False, ones, G, (125, 1, 32))
    assert float((dk - W.grad).abs().max()) <= 2e-5 * float(dk.abs().max()), float((dk - W.grad).abs().max())


@pytest.mark.parametrize("n_pts", [37, 101])
def test_input_layer_weight_gradient_tail_tile(n_pts):
    """conv0_wgrad_unit_kernel on maps smaller than one 128-voxel workgroup tile and not a multiple of 32 (tail tile, idle-wave
    partials): the unit-feature MFMA path against the explicit-feature fp32 path AND a host fp64 sum over the kernel map."""
    from egonn_amd import _lib
    dev = _lib.require_gpu()
    ctx = _lib.Context(dev, coord_bits=12)
    g = np.random.default_rng(n_pts)
    pts = (g.integers(-6, 7, size=(n_pts, 3)).astype(np.float32) + 0.5) * 0.2        # a dense little cluster: many k=5 neighbours
    ctx.voxelize(torch.from_numpy(pts).to(dev), [0, n_pts], 0, [0.2])
    n0 = ctx.level_count(0)
    assert 0 < n0 < 128
    G = rnd((n0, 32), 6, dev)
    dk_unit = ctx.conv_backward_weight(0, 0, 5, False, None, G, (125, 1, 32))
    dk_feat = ctx.conv_backward_weight(0, 0, 5, False, torch.ones((n0, 1), device=dev), G, (125, 1, 32))
    coords = ctx.level_coords(0).cpu().numpy()
    ref = np.zeros((125, 32))
    Gn = G.double().cpu().numpy()
    lut = {tuple(c): i for i, c in enumerate(coords.tolist())}
    for o, c in enumerate(coords.tolist()):
        for k in range(125):       # ME kernel index: first spatial axis fastest
            d = (k % 5 - 2, (k // 5) % 5 - 2, k // 25 - 2)
            j = lut.get((c[0], c[1] + d[0], c[2] + d[1], c[3] + d[2]))
            if j is not None:
                ref[k] += Gn[o]
    for dk in (dk_unit, dk_feat):
        err = np.abs(dk.double().cpu().numpy().reshape(125, 32) - ref).max()
        assert err <= 2e-5 * np.abs(ref).max(), err


@pytest.mark.parametrize("relu", [True, False])
def test_batch_norm_train_matches_torch(plan, relu):
    from egonn_amd.train import BatchNormFn
    ctx, dev = plan, plan.device
    n, c = ctx.level_count(2), 64
    x = (rnd((n, c), 8, dev) * 1.7 + 0.9).requires_grad_(True)
    bn = torch.nn.BatchNorm1d(c).to(dev)
    with torch.no_grad():
        bn.weight.copy_(rnd((c,), 9, dev) * 0.2 + 1.0)
        bn.bias.copy_(rnd((c,), 10, dev) * 0.1)
    G = rnd((n, c), 11, dev)
    y = BatchNormFn.apply(x, bn.weight, bn.bias, ctx, bn, relu, None, None)
    (y * G).sum().backward()
    ref = torch.nn.BatchNorm1d(c)
    with torch.no_grad():
        ref.weight.copy_(bn.weight.cpu())
        ref.bias.copy_(bn.bias.cpu())
    xr = x.detach().cpu().double().requires_grad_(True)
    ref = ref.double()
    yr = ref(xr)
    if relu:
        yr = torch.relu(yr)
    (yr * G.cpu().double()).sum().backward()
    assert torch.allclose(y.detach().cpu().double(), yr.detach(), rtol=1e-4, atol=1e-5)
    assert torch.allclose(x.grad.cpu().double(), xr.grad, rtol=1e-3, atol=1e-5)
    assert torch.allclose(bn.weight.grad.cpu().double(), ref.weight.grad, rtol=1e-4, atol=1e-3)
    assert torch.allclose(bn.bias.grad.cpu().double(), ref.bias.grad, rtol=1e-4, atol=1e-3)
    assert torch.allclose(bn.running_mean.cpu().double(), ref.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn.running_var.cpu().double(), ref.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == 1


def test_eca_tail_gem_linear_match_torch(plan):
    from egonn_amd import train as T
    from egonn_amd.model import ECALayer
    ctx, dev = plan, plan.device
    level, c = 3, 64
    n, B = ctx.level_count(level), ctx.batch_size
    off = ctx.level_batch_offsets(level)
    x = rnd((n, c), 12, dev).requires_grad_(True)
    res = rnd((n, c), 13, dev).requires_grad_(True)
    eca = ECALayer(c).to(dev)
    lin = torch.nn.Linear(c, 128).to(dev)
    p = torch.nn.Parameter(torch.tensor([3.0], device=dev))
    G = rnd((B, 128), 14, dev)
    h = T.eca_tail(ctx, level, x, res, eca)
    h = T.LinearFn.apply(h, lin.weight, lin.bias, ctx, True)
    g = T.GeMFn.apply(h, p, ctx, level)
    (g * G).sum().backward()

    # torch (CPU, fp64) restatement of the same three reference modules
    xr, rr = x.detach().cpu().double().requires_grad_(True), res.detach().cpu().double().requires_grad_(True)
    wr = eca.conv.weight.detach().cpu().double().requires_grad_(True)
    lw, lb = lin.weight.detach().cpu().double().requires_grad_(True), lin.bias.detach().cpu().double().requires_grad_(True)
    pr = p.detach().cpu().double().requires_grad_(True)
    outs = []
    for b in range(B):
        xb, rb = xr[off[b]:off[b + 1]], rr[off[b]:off[b + 1]]
        m = xb.mean(0, keepdim=True)
        gate = torch.sigmoid(torch.nn.functional.conv1d(m.unsqueeze(1), wr, padding=(wr.shape[-1] - 1) // 2).squeeze(1))
        hb = torch.relu(xb * gate + rb)
        hb = torch.relu(hb @ lw.t() + lb)
        outs.append(hb.clamp(min=1e-6).pow(pr).mean(0).pow(1.0 / pr))
    gr = torch.stack(outs)
    (gr * G.cpu().double()).sum().backward()
    assert torch.allclose(g.detach().cpu().double(), gr.detach(), rtol=1e-4, atol=1e-6)
    for mine, ref, name in [(x.grad, xr.grad, "x"), (res.grad, rr.grad, "res"), (eca.conv.weight.grad, wr.grad, "eca"),
                            (lin.weight.grad, lw.grad, "lin.w"), (lin.bias.grad, lb.grad, "lin.b"), (p.grad, pr.grad, "p")]:
        scale = float(ref.abs().max())
        assert torch.allclose(mine.cpu().double(), ref, rtol=1e-3, atol=1e-4 * scale + 1e-9), name


def _digest(name, g):
    g = np.asarray(g, dtype=np.float64).reshape(-1)
    r = np.random.default_rng(zlib.crc32(name.encode())).standard_normal(g.size)
    return np.concatenate([[np.linalg.norm(g), float(g @ r)], g[:64] if g.size > 4096 else g])


def _row_weights(coords, width, salt):
    c = np.asarray(coords, dtype=np.float64)
    phase = 2.1 * c[:, 0] + 0.37 * c[:, 1] + 0.73 * c[:, 2] + 1.13 * c[:, 3] + salt
    return np.cos(phase[:, None] + 0.05 * np.arange(width)[None, :]).astype(np.float32)


@pytest.mark.parametrize("name", ["egonn_train_cart03", "egonn_train_polar"])
def test_train_step_matches_reference_fixture(name):
    """forward (batch-statistics BN, all four outputs), backward into all 104 parameters and the running-stat update
    of one step vs the autograd of the reference's own graph (tests/golden/make_golden.py train)."""
    import __graft_entry__ as ge
    ge.build()
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
    model = model.to(dev).train()
    coords = torch.from_numpy(case["coords"]).to(dev)
    y = model({"coords": coords, "features": torch.ones((len(coords), 1), device=dev)})
    g = y["global"]
    ref_g = case["global"]
    assert g.shape == ref_g.shape
    assert H.cosine_err(g.detach().cpu().numpy(), ref_g).max() <= 1e-4
    assert np.allclose(g.detach().cpu().numpy(), ref_g, rtol=2e-3, atol=2e-4 * np.abs(ref_g).max())
    R = torch.from_numpy(np.random.default_rng(int(case["proj_seed"])).standard_normal(ref_g.shape).astype(np.float32)).to(dev)
    loss = (g * R).sum()
    kcs = model.keypoint_coords()
    for b in range(int(case["n_scans"])):
        kc = kcs[b].cpu().numpy()
        perm = H.join_perm(kc, case[f"kp_coords_{b}"])              # fixture row -> my row
        d, k, sg = y["descriptors"][b], y["keypoints"][b], y["sigma"][b]
        assert H.cosine_err(d.detach().cpu().numpy()[perm], case[f"descriptors_{b}"]).max() <= 1e-4
        assert np.allclose(k.detach().cpu().numpy()[perm], case[f"keypoints_{b}"], atol=2e-3)
        assert np.allclose(sg.detach().cpu().numpy()[perm], case[f"sigma_{b}"], rtol=2e-3, atol=1e-5)
        loss = loss + (d * torch.from_numpy(_row_weights(kc, 128, 0.1)).to(dev)).sum() \
                    + (k * torch.from_numpy(_row_weights(kc, 3, 0.2)).to(dev)).sum() \
                    + (sg * torch.from_numpy(_row_weights(kc, 1, 0.3)).to(dev)).sum()
    assert abs(loss.item() - float(case["loss"])) <= 2e-3 * max(1.0, abs(float(case["loss"])))
    loss.backward()
    grads = {k: p.grad for k, p in model.named_parameters()}
    keys = [k[5:] for k in case if k.startswith("grad/")]
    assert len(keys) == 104 == len(grads)

    def digest_err(mine, ref):
        norm = max(ref[0], 1e-12)
        return max(abs(mine[0] - ref[0]) / norm, abs(mine[1] - ref[1]) / norm,
                   float(np.abs(mine[2:] - ref[2:]).max()) / max(float(np.abs(ref[2:]).max()), 1e-12))

    # How well is this gradient DEFINED in fp32?  The same step on the plain one-thread-per-output kernels and on the exact fp32 MFMA
    # kernels — two exact-fp32 summation orders of the same convolutions.  With the local-head terms in the loss the two disagree by
    # up to 2e-2 on the first layers of the polar case (measured, tools/exp/r06_train_grad_sensitivity.py: a perturbation of 1e-7 at
    # level 3 moves those entries by percent, whichever kernel causes it), so an entry is held to the reference within
    # max(5e-3, 2 x that spread): 5e-3 wherever fp32 pins the value, the fp32 noise floor of the quantity where it does not.
    def grads_under(setting):
        m2 = egonn_amd.model_factory(mp)
        m2.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
        m2 = m2.to(dev).train()
        c2 = m2.context()
        if setting == "plain":
            c2.set_naive_conv(True)
        else:
            c2.set_exact_fp32(True)
        y2 = m2({"coords": coords, "features": torch.ones((len(coords), 1), device=dev)})
        l2 = (y2["global"] * R).sum()
        k2 = m2.keypoint_coords()
        for b in range(int(case["n_scans"])):
            kc = k2[b].cpu().numpy()
            l2 = l2 + (y2["descriptors"][b] * torch.from_numpy(_row_weights(kc, 128, 0.1)).to(dev)).sum() \
                    + (y2["keypoints"][b] * torch.from_numpy(_row_weights(kc, 3, 0.2)).to(dev)).sum() \
                    + (y2["sigma"][b] * torch.from_numpy(_row_weights(kc, 1, 0.3)).to(dev)).sum()
        l2.backward()
        return {k: _digest(k, p.grad.detach().cpu().numpy()) for k, p in m2.named_parameters()}
    g_plain, g_exact = grads_under("plain"), grads_under("exact")
    bad = []
    for k in keys:
        assert grads[k] is not None, k
        mine, ref = _digest(k, grads[k].detach().cpu().numpy()), case["grad/" + k]
        err = digest_err(mine, ref)
        floor = digest_err(g_exact[k], g_plain[k])
        if err > max(5e-3, 2.0 * floor):
            bad.append((k, err, floor))
    assert not bad, bad
    sd = model.state_dict()
    for k in [k[4:] for k in case if k.startswith("buf/")]:
        assert np.allclose(sd[k].cpu().numpy(), case["buf/" + k], rtol=1e-3, atol=1e-5), k


# ----------------------------------------------------------------------------- sharded step == single-process step
def _make_model(dev, wseed):
    import egonn_amd
    mp_ = egonn_amd.ModelParams(model="egonn", coordinates="cartesian", quantization_step=0.3)
    model = egonn_amd.model_factory(mp_)
    w = H.seeded_weights(wseed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    return model.to(dev)


def _scan_batch(dev, coords, scan_ids):
    """sub-batch of the fixture's scans, batch index renumbered from 0"""
    parts = []
    for new_b, b in enumerate(scan_ids):
        c = coords[coords[:, 0] == b].clone()
        c[:, 0] = new_b
        parts.append(c)
    c = torch.cat(parts).to(dev)
    return {"coords": c, "features": torch.ones((len(c), 1), device=dev), "batch_size": len(scan_ids)}


def _masks():
    pos = torch.zeros((3, 3), dtype=torch.bool)
    pos[0, 1] = pos[1, 0] = True
    neg = torch.zeros((3, 3), dtype=torch.bool)
    neg[0, 2] = neg[2, 0] = neg[1, 2] = neg[2, 1] = True
    return pos, neg


def _sharded_worker(rank, world, port, out_path, backend="gloo"):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # gloo: 2 ranks sharing the one GPU of the box (host-staged collectives); nccl (= RCCL): one GPU per rank
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from egonn_amd.train import TrainStep
        case = H.load_case("egonn_train_cart03")
        coords = torch.from_numpy(case["coords"])
        model = _make_model(dev, int(case["weight_seed"]))
        step = TrainStep(model, torch.optim.SGD(model.parameters(), lr=0.0), margin=0.2)
        pos, neg = _masks()
        mine = [0, 1] if rank == 0 else [2]                            # uneven shards
        from egonn_amd import distributed as D
        D.COLLECTIVES.clear()
        loss, stats = step(_scan_batch(dev, coords, mine), pos, neg, step_optimizer=False, shard_sizes=[2, 1])
        comm = dict(D.COLLECTIVES)
        n_bn = sum(1 for m in model.modules() if isinstance(m, torch.nn.BatchNorm1d) and m.weight.grad is not None)
        grads = {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None}
        bufs = {k: v.detach().cpu() for k, v in model.state_dict().items() if "running" in k}
        torch.save({"loss": float(loss), "stats": stats, "grads": grads, "bufs": bufs, "comm": comm, "n_bn": n_bn}, f"{out_path}.{rank}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_sharded_step_equals_single_process_step(tmp_path, backend):
    """2 ranks (scans [0,1] | [2]) with SyncBN + embedding all-gather + gradient all-reduce reproduce the gradients,
    the loss and the BatchNorm running statistics of one process stepping the whole batch.  backend nccl = RCCL with one
    GPU per rank: runs on the first box with >= 2 GPUs (skipped on the 1-GPU boxes)."""
    import socket
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one GPU per rank: >= 2 GPUs")
    import torch.multiprocessing as tmp_mp
    import __graft_entry__ as ge
    ge.build()
    from egonn_amd import _lib
    from egonn_amd.train import TrainStep
    dev = _lib.require_gpu()
    case = H.load_case("egonn_train_cart03")
    coords = torch.from_numpy(case["coords"])
    model = _make_model(dev, int(case["weight_seed"]))
    step = TrainStep(model, torch.optim.SGD(model.parameters(), lr=0.0), margin=0.2)
    pos, neg = _masks()
    loss, stats = step(_scan_batch(dev, coords, [0, 1, 2]), pos, neg, step_optimizer=False)
    assert stats["num_triplets"] == 2 and np.isfinite(float(loss))
    want = {k: p.grad.detach().cpu() for k, p in model.named_parameters() if p.grad is not None}
    want_buf = {k: v.detach().cpu() for k, v in model.state_dict().items() if "running" in k}

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "res")
    ctx = tmp_mp.get_context("spawn")
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, out, backend)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    for r in range(2):
        got = torch.load(f"{out}.{r}")
        # the communication pattern of one step: the level row totals (1 all-reduce of 8 values), ONE all-reduce of the (2, C)
        # sums per BatchNorm and direction (SyncBN), ONE all-gather of the embeddings, ONE flat all-reduce of the gradients
        assert got["n_bn"] >= 20
        assert got["comm"] == {"all_reduce": 1 + 2 * got["n_bn"] + 1, "all_gather": 1}, got["comm"]
        assert abs(got["loss"] - float(loss)) <= 1e-4 * max(1.0, abs(float(loss)))
        assert set(got["grads"]) == set(want)
        for k, g in got["grads"].items():
            scale = float(want[k].abs().max())
            assert torch.allclose(g, want[k], rtol=2e-3, atol=2e-4 * scale + 1e-10), (r, k)
        for k, v in got["bufs"].items():
            if k.startswith(("trunk", "global")):
                assert torch.allclose(v, want_buf[k], rtol=1e-4, atol=1e-6), (r, k)


def test_train_step_is_bitwise_deterministic():
    """two steps from the same state give bitwise identical loss and gradients (fixed-order reductions everywhere; the
    ECA Conv1d is written as shifted multiply-adds because F.conv1d's weight gradient uses atomics on this backend)."""
    import __graft_entry__ as ge
    ge.build()
    from egonn_amd import _lib
    from egonn_amd.train import TrainStep
    dev = _lib.require_gpu()
    case = H.load_case("egonn_train_cart03")
    coords = torch.from_numpy(case["coords"])
    pos, neg = _masks()
    runs = []
    for _ in range(2):
        model = _make_model(dev, int(case["weight_seed"]))
        step = TrainStep(model, torch.optim.SGD(model.parameters(), lr=0.0), margin=0.2)
        loss, _ = step(_scan_batch(dev, coords, [0, 1, 2]), pos, neg, step_optimizer=False)
        runs.append((float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}))
    assert runs[0][0] == runs[1][0]
    for k, g in runs[0][1].items():
        assert torch.equal(g, runs[1][1][k]), k


@pytest.mark.parametrize("name,shapes", [("minkloc3d_train_cart03", "minkloc3d_cart03_b2"),
                                         ("minkloc_eca_train_cart03", "minkloc_eca_cart03")])
def test_minkloc_train_step_matches_reference_fixture(name, shapes):
    """MinkLoc3D / MinkLoc (MinkFPN + GeM; BasicBlock and ECABasicBlock) in train mode vs the reference graph's autograd."""
    import __graft_entry__ as ge
    ge.build()
    import egonn_amd
    from egonn_amd import _lib
    dev = _lib.require_gpu()
    case = H.load_case(name)
    if str(case["model"]) == "MinkLoc3D":
        model = egonn_amd.model_factory(egonn_amd.ModelParams(model="MinkLoc3D", coordinates="cartesian", quantization_step=0.3))
    else:
        model = egonn_amd.model_factory(egonn_amd.ModelParams(model="MinkLoc", coordinates="cartesian", quantization_step=0.3,
                                                             block="ECABasicBlock", planes="32,64,64", layers="1,1,1"))
    w = H.seeded_weights(int(case["weight_seed"]), shapes)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    model = model.to(dev).train()
    coords = torch.from_numpy(case["coords"]).to(dev)
    g = model({"coords": coords, "features": torch.ones((len(coords), 1), device=dev)})["global"]
    assert H.cosine_err(g.detach().cpu().numpy(), case["global"]).max() <= 1e-4
    R = torch.from_numpy(np.random.default_rng(int(case["proj_seed"])).standard_normal(case["global"].shape).astype(np.float32)).to(dev)
    loss = (g * R).sum()
    assert abs(loss.item() - float(case["loss"])) <= 2e-3 * max(1.0, abs(float(case["loss"])))
    loss.backward()
    grads = {k: p.grad for k, p in model.named_parameters()}
    keys = [k[5:] for k in case if k.startswith("grad/")]
    assert set(keys) == set(grads)
    bad = []
    for k in keys:
        assert grads[k] is not None, k
        mine, ref = _digest(k, grads[k].detach().cpu().numpy()), case["grad/" + k]
        norm = max(ref[0], 1e-12)
        err = max(abs(mine[0] - ref[0]) / norm, abs(mine[1] - ref[1]) / norm,
                  float(np.abs(mine[2:] - ref[2:]).max()) / max(float(np.abs(ref[2:]).max()), 1e-12))
        if err > 5e-3:
            bad.append((k, err))
    assert not bad, bad
    sd = model.state_dict()
    for k in [k[4:] for k in case if k.startswith("buf/")]:
        assert np.allclose(sd[k].cpu().numpy(), case["buf/" + k], rtol=1e-3, atol=1e-5), k


@pytest.mark.parametrize("cin", [32, 64, 128, 192, 256])
def test_dense_every_kernel_path_matches_fp64(plan, cin):
    """egonn_dense (MinkowskiLinear / 1x1 convolution, models/minkgl.py:175-225) on every row-count regime of its three
    kernels — dense_small (< 8192 rows, Cin <= 128), dense_lds (weights staged in LDS; column blocks over grid.y when the
    weight set exceeds 96 KB: Cin*Cout*4 > 96 KB) and the 64-column kernel — both weight layouts, bias, ReLU, ragged row
    counts and column counts that are not multiples of 16, against an fp64 matmul.  fp32 MFMA accumulation over <= 256
    terms: |err| <= 2e-5 * sum|x||w| is generous."""
    ctx, dev = plan, plan.device
    seed = 1000 + cin
    for n in (1, 37, 2047, 2048, 8191, 8192, 8200, 33000):
        for cout, out_in, use_bias, act in ((16, 1, False, 0), (64, 0, True, 1), (100, 1, True, 0), (192, 0, False, 0),
                                             (256, 1, True, 1), (256, 0, False, 0), (3, 1, True, 0)):
            if n > 9000 and cout in (100, 3):
                continue
            seed += 1
            x = rnd((n, cin), seed, dev)
            w = rnd((cout, cin) if out_in else (cin, cout), seed + 7, dev, 0.2)
            b = rnd((cout,), seed + 9, dev) if use_bias else None
            got = ctx.dense(x, w, bool(out_in), b, act).cpu().double()
            wd = w.cpu().double()
            wd = wd.t() if out_in else wd
            want = x.cpu().double() @ wd
            bound = 2e-5 * (x.cpu().double().abs() @ wd.abs()) + 1e-6
            if use_bias:
                want = want + b.cpu().double()
            if act == 1:
                want = torch.relu(want)
            err = (got - want).abs()
            assert bool((err <= bound).all()), (n, cin, cout, out_in, float(err.max()), float(bound.min()))
