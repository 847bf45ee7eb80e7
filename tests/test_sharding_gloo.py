"""N>1 path on CPU: world_size-2 gloo processes shard a batch, solve their shard and gather.

The product's solver needs a GPU, so here the local solve is the CPU oracle (tests may use
it as the checker); what is under test is the host logic of diff-qp-mpc_amd/sharding.py:
partition bounds (incl. ragged), replication of shared parameters, the single all_gather,
and the mean-over-full-batch reduction of shared-parameter gradients."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, B, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diff_qp_mpc_amd import sharding
    from oracle import oracle
    g = torch.Generator().manual_seed(5)
    nz, nineq, neq = 8, 6, 3
    L = torch.randn(nz, nz, generator=g, dtype=torch.float64)
    Q = L @ L.T + 1e-3 * torch.eye(nz, dtype=torch.float64)          # shared (no batch dim)
    G = torch.randn(B, nineq, nz, generator=g, dtype=torch.float64)
    z0 = torch.randn(B, nz, generator=g, dtype=torch.float64)
    h = (G @ z0.unsqueeze(-1)).squeeze(-1) + torch.rand(B, nineq, generator=g, dtype=torch.float64)
    A = torch.randn(B, neq, nz, generator=g, dtype=torch.float64)
    b = (A @ z0.unsqueeze(-1)).squeeze(-1)
    p = torch.randn(B, nz, generator=g, dtype=torch.float64)

    store = {}

    def solve(Ql, pl, Gl, hl, Al, bl):
        Bl = pl.shape[0]
        o = oracle.qp_forward(oracle.expand(Ql.numpy(), Bl, 3), pl.numpy(), Gl.numpy(),
                              hl.numpy(), Al.numpy(), bl.numpy(), nthreads=1)
        gr = oracle.qp_backward(oracle.expand(Ql.numpy(), Bl, 3), Gl.numpy(), Al.numpy(),
                                o["zhat"], o["lam"], o["nu"], o["slack"], np.ones((Bl, nz)),
                                nthreads=1)
        store["dQ_sum"] = torch.tensor(gr["dQ"].sum(0))
        return torch.tensor(o["zhat"])

    z, (lo, hi) = sharding.solve_sharded(solve, (Q, p, G, h, A, b))
    dQ = sharding.reduce_shared_grad(store["dQ_sum"], B)
    if rank == 0:
        q.put((z.numpy(), dQ.numpy(), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 7])
def test_two_rank_shard_solve_gather(B):
    from oracle import oracle
    from diff_qp_mpc_amd import sharding
    assert sharding.shard_bounds(7, 2, 0) == (0, 4) and sharding.shard_bounds(7, 2, 1) == (4, 7)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 200) + B
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    z, dQ, (lo, hi) = q.get(timeout=120)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    # single-process answer on the full batch
    g = torch.Generator().manual_seed(5)
    nz, nineq, neq = 8, 6, 3
    L = torch.randn(nz, nz, generator=g, dtype=torch.float64)
    Q = (L @ L.T + 1e-3 * torch.eye(nz, dtype=torch.float64)).numpy()
    G = torch.randn(B, nineq, nz, generator=g, dtype=torch.float64)
    z0 = torch.randn(B, nz, generator=g, dtype=torch.float64)
    h = (G @ z0.unsqueeze(-1)).squeeze(-1) + torch.rand(B, nineq, generator=g, dtype=torch.float64)
    A = torch.randn(B, neq, nz, generator=g, dtype=torch.float64)
    b = (A @ z0.unsqueeze(-1)).squeeze(-1)
    p = torch.randn(B, nz, generator=g, dtype=torch.float64)
    Qe = oracle.expand(Q, B, 3)
    o = oracle.qp_forward(Qe, p.numpy(), G.numpy(), h.numpy(), A.numpy(), b.numpy())
    gr = oracle.qp_backward(Qe, G.numpy(), A.numpy(), o["zhat"], o["lam"], o["nu"], o["slack"],
                            np.ones((B, nz)))
    assert z.shape == (B, nz) and lo == 0
    # the oracle's termination is batch-coupled, so shards may stop at different iterations
    np.testing.assert_allclose(z, o["zhat"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(dQ, gr["dQ"].mean(0), rtol=1e-6, atol=1e-9)


def _worker_autograd(rank, world, port, B, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diff_qp_mpc_amd import sharding, policies
    lo, hi = sharding.shard_bounds(B, world, rank)
    g = torch.Generator().manual_seed(3)
    zfull = torch.randn(B, 4, generator=g, dtype=torch.float64)
    w = torch.randn(B, 4, generator=g, dtype=torch.float64)
    z_local = zfull[lo:hi].clone().requires_grad_()
    z = sharding.gather_solution_autograd(z_local * 2.0, B)         # differentiable gather
    (z * w).sum().backward()
    # shared-parameter gradient: every rank holds the LOCAL mean, shards are ragged for B = 7
    per_sample = torch.arange(B, dtype=torch.float64)[:, None] * torch.ones(B, 3, dtype=torch.float64)
    g_mean = sharding.reduce_shared_grad_from_local_mean(per_sample[lo:hi].mean(0), hi - lo, B)
    # data-parallel DEQLayer gradients: one flat all_reduce, averaged over ranks
    lin = torch.nn.Linear(3, 2)
    with torch.no_grad():
        lin.weight.fill_(0.5); lin.bias.fill_(0.1)
    x = torch.full((2, 3), float(rank + 1))
    lin(x).sum().backward()
    n = policies.allreduce_gradients(lin)
    if rank == 0:
        q.put((z.detach().numpy(), z_local.grad.numpy(), g_mean.numpy(), lin.weight.grad.numpy(), n, (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 7])
def test_autograd_gather_ragged_mean_and_flat_gradient_allreduce(B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29850 + (os.getpid() % 100) + B
    procs = [ctx.Process(target=_worker_autograd, args=(r, 2, port, B, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    z, gz, g_mean, gw, n, (lo, hi) = q.get(timeout=120)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    g = torch.Generator().manual_seed(3)
    zfull = torch.randn(B, 4, generator=g, dtype=torch.float64)
    w = torch.randn(B, 4, generator=g, dtype=torch.float64)
    np.testing.assert_allclose(z, 2.0 * zfull.numpy())
    np.testing.assert_allclose(gz, 2.0 * w.numpy()[lo:hi])                     # the cotangent of the own shard
    np.testing.assert_allclose(g_mean, np.full(3, (B - 1) / 2.0))              # mean over ALL samples, ragged or not
    np.testing.assert_allclose(gw, np.full((2, 3), 2 * 1.5))                   # (2*1 + 2*2) / 2 ranks
    assert n == 8
