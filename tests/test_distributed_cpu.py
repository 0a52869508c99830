"""World-size-2 `gloo` tests (CPU) of the multi-GPU database build: sharding + the single all-gather of the
global descriptors.  The HIP extractor is replaced by a deterministic stand-in, so this covers the N>1 host
logic that the driver's 8-GPU run exercises over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from egonn_amd.distributed import DatabaseBuilder, all_gather_rows, shard_bounds


class FakeExtractor:
    """global descriptor = a fixed function of the scan, so the gathered matrix is checkable."""

    def extract(self, scans):
        g = torch.stack([torch.cat([s.sum(0), s.mean(0), torch.tensor([float(len(s))])]) for s in scans])
        b = len(scans)
        return {"global": g, "keypoints": torch.zeros((b, 4, 3)), "descriptors": torch.zeros((b, 4, 8)),
                "count": torch.full((b,), 4, dtype=torch.int32)}


def load_scan(i):
    rng = np.random.default_rng(i)
    return torch.from_numpy(rng.standard_normal((10 + i % 7, 3)).astype(np.float32))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_scans, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = DatabaseBuilder(FakeExtractor(), batch_size=4).build(load_scan, n_scans)
        q.put((rank, res["global"].numpy(), res["range"], int(res["count"].shape[0])))
    finally:
        dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 16, 20000):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("n_scans", [9, 16])          # uneven and even shards
def test_database_build_world2_gloo(n_scans):
    want = DatabaseBuilder(FakeExtractor(), batch_size=4).build(load_scan, n_scans)["global"].numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_scans, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ranges = {}
    for rank, g, rng, n_local in got:
        assert g.shape == want.shape and np.array_equal(g, want)      # same matrix, scan order preserved, every rank
        ranges[rank] = rng
        assert n_local == rng[1] - rng[0]                             # local outputs stay rank-local
    assert ranges[0][1] == ranges[1][0] and ranges[1][1] == n_scans


def test_all_gather_rows_single_process():
    x = torch.arange(12.0).reshape(4, 3)
    assert torch.equal(all_gather_rows(x, 4), x)


def _gather_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from egonn_amd.distributed import all_gather_embeddings
        torch.manual_seed(0)
        full = torch.randn(7, 5)
        lo, hi = (0, 4) if rank == 0 else (4, 7)                     # ragged shards
        local = full[lo:hi].clone().requires_grad_(True)
        g = all_gather_embeddings(local)
        w = torch.arange(35.0).reshape(7, 5)
        (g * w).sum().backward()                                      # same "loss" on every rank
        # shard sizes known to every rank (the sampler's partition): ONE collective, no size exchange
        local2 = full[lo:hi].clone().requires_grad_(True)
        g2 = all_gather_embeddings(local2, sizes=[4, 3])
        (g2 * w).sum().backward()
        assert torch.equal(g2, g) and torch.equal(local2.grad, local.grad)
        try:
            all_gather_embeddings(local2, sizes=[3, 4] if rank == 0 else [4, 4])
            raise AssertionError("mismatching shard sizes must be rejected")
        except ValueError:
            pass
        q.put((rank, g.detach().numpy(), local.grad.numpy(), w[lo:hi].numpy(), full.numpy()))
    finally:
        dist.destroy_process_group()


def test_all_gather_embeddings_autograd_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, g, grad, want_grad, full in got:
        assert np.array_equal(g, full)                 # every rank sees the whole batch, rank order
        assert np.array_equal(grad, want_grad)         # and back-propagates exactly its own rows


# ----------------------------------------------------------------------------- sharded training step: host logic
def _stats_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from egonn_amd.train import combine_batch_stats, all_reduce_gradients
        g = torch.Generator().manual_seed(5)
        x = torch.randn((101, 7), generator=g) * 2 + 3
        lo, hi = (0, 40) if rank == 0 else (40, 101)                     # uneven shards
        mean, total = combine_batch_stats(x[lo:hi].sum(0), torch.tensor(float(hi - lo)), dist.group.WORLD)
        # a "parameter" whose per-rank gradient is the contribution of the rank's rows
        p = torch.nn.Parameter(torch.zeros(7))
        p.grad = x[lo:hi].sum(0)
        q2 = torch.nn.Parameter(torch.zeros(3))                           # no gradient on any rank: must be skipped
        all_reduce_gradients([p, q2])
        q.put((rank, mean.numpy(), float(total), p.grad.numpy(), q2.grad is None))
    finally:
        dist.destroy_process_group()


def test_syncbn_statistics_and_gradient_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stats_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(5)
    x = torch.randn((101, 7), generator=g) * 2 + 3
    for rank, mean, total, grad, skipped in got:
        assert total == 101.0 and skipped
        assert np.allclose(mean, x.mean(0).numpy(), rtol=1e-6, atol=1e-6)          # whole-batch mean on every rank
        assert np.allclose(grad, x.sum(0).numpy(), rtol=1e-6, atol=1e-5)           # SUM, not average


def test_bench_distributed_branch_dry_run_world2():
    """bench.py's distributed branch exactly as the driver launches it (torch.distributed.run, one rank per GPU, rendezvous on
    127.0.0.1), with the gloo backend and a stub step: env parsing, process group, the barrier-bracketed timed regions with the
    max-over-ranks reduction, the ranks-seen all-reduce and rank 0's single JSON line."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--repeats", "2", "--dry-run"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, cwd=repo)
    assert r.returncode == 0, r.stderr[-800:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # rank 0 only
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["value"] is None    # never mistaken for a measurement
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["repeats"]["timed_regions"] == 2 and d["ms_per_step"] > 0
