#!/usr/bin/env python3
"""qp_wrapper.MPC single-QP call (n=3 m=3 T=5, B=4096, LinDx, box bounds) eager vs captured in a
hipGraph (torch.cuda.make_graphed_callables over forward AND backward): the C-ABI entry points only
enqueue on the current stream, allocate nothing and never synchronise, so the whole call replays
as one graph launch."""
import os, sys, time, faulthandler; faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff_qp_mpc_amd.qp_wrapper import MPC, QuadCost, LinDx, graphed_mpc

n, m, T = 3, 3, 5
B = int(os.environ.get("BATCH", "4096"))
gen = torch.Generator().manual_seed(42)
Ad = torch.eye(n, dtype=torch.float64) + 0.2 * torch.randn(n, n, generator=gen, dtype=torch.float64)
Bd = torch.randn(n, m, generator=gen, dtype=torch.float64)
mk = lambda t: t.cuda().requires_grad_()
C = mk(torch.eye(n + m, dtype=torch.float64).repeat(T, B, 1, 1))
c = mk(torch.randn(T, B, n + m, generator=gen, dtype=torch.float64))
x0 = mk(torch.randn(B, n, generator=gen, dtype=torch.float64))
F = mk(torch.cat([Ad, Bd], 1).repeat(T - 1, B, 1, 1))
f = mk(torch.zeros(T - 1, B, n, dtype=torch.float64))
one = torch.ones(m, dtype=torch.float64).cuda()
mpc = MPC(n, m, T, u_lower=-one, u_upper=one, n_batch=B, verbose=-1, single_qp_solve=True)


def eager(x0, C, c, F, f):
    return mpc(x0, QuadCost(C, c), LinDx(F, f), None)


def timeit(fn, reps=50):
    for _ in range(5):
        x, u = fn(x0, C, c, F, f); (x.sum() + 2.0 * u.sum()).backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        x, u = fn(x0, C, c, F, f); (x.sum() + 2.0 * u.sum()).backward()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


t_eager = timeit(eager)
xe, ue = eager(x0, C, c, F, f)
g = graphed_mpc(mpc, (x0, C, c, F, f))
t_graph = timeit(g)
xg, ug = g(x0, C, c, F, f)
print("B=%d  eager %.3f ms per call (fwd+bwd)   hipGraph %.3f ms per call   max |dx| %.1e max |du| %.1e" % (
    B, t_eager, t_graph, float((xe - xg).abs().max()), float((ue - ug).abs().max())))
print("trajectories/s: eager %.2f M, graph %.2f M" % (B / t_eager / 1e3, B / t_graph / 1e3))
