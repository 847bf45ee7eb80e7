#!/usr/bin/env python3
"""End-to-end timing of the qp_wrapper.MPC mirror (rows a12-a13) at the metric shape
n=3 m=3 T=5, B=4096, LinDx, box |u| <= 1: forward (assembly kernel + QP forward + rollout/cost
bookkeeping in torch) and backward (QP backward + assembly adjoint), with a torch profiler
table of where the time goes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff_qp_mpc_amd.qp_wrapper import MPC, QuadCost, LinDx

n, m, T, B = 3, 3, 5, 4096
gen = torch.Generator().manual_seed(42)
Ad = torch.eye(n, dtype=torch.float64) + 0.2 * torch.randn(n, n, generator=gen, dtype=torch.float64)
Bd = torch.randn(n, m, generator=gen, dtype=torch.float64)
C = torch.eye(n + m, dtype=torch.float64).repeat(T, B, 1, 1).cuda().requires_grad_()
c = torch.randn(T, B, n + m, generator=gen, dtype=torch.float64).cuda().requires_grad_()
x0 = torch.randn(B, n, generator=gen, dtype=torch.float64).cuda().requires_grad_()
F = torch.cat([Ad, Bd], 1).repeat(T - 1, B, 1, 1).cuda().requires_grad_()
f = torch.zeros(T - 1, B, n, dtype=torch.float64).cuda().requires_grad_()
one = torch.ones(m, dtype=torch.float64).cuda()
for mode, kw in (("single_qp_solve", dict(single_qp_solve=True)), ("sqp qp_iter=3", dict(qp_iter=3))):
    mpc = MPC(n, m, T, u_lower=-one, u_upper=one, n_batch=B, verbose=-1, **kw)
    def step():
        x, u = mpc(x0, QuadCost(C, c), LinDx(F, f), None)
        (x.sum() + 2.0 * u.sum()).backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print("%-18s forward+backward %.3f ms per MPC call  (%.2f M trajectories/s)" % (mode, dt * 1e3, B / dt / 1e6), flush=True)
    if mode.startswith("single"):
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            for _ in range(3):
                step()
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))

if os.environ.get("CPROFILE"):
    import cProfile, pstats
    mpc = MPC(n, m, T, u_lower=-one, u_upper=one, n_batch=B, verbose=-1, single_qp_solve=True)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5):
        x, u = mpc(x0, QuadCost(C, c), LinDx(F, f), None)
        (x.sum() + 2.0 * u.sum()).backward()
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
