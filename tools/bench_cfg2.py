#!/usr/bin/env python3
"""BASELINE config 2 shape end to end: qp_wrapper.MPC on the PendulumDx device model (n 3, m 1, T 10),
B = 1024, single-QP mode and SQP (qp_iter 3), forward + backward; the QP solve on the stage-wise kernels
(true-dynamics residual in the iterations) vs the dense one-QP-per-wavefront kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diff_qp_mpc_amd import qp_wrapper
from diff_qp_mpc_amd.dynamics import DeviceDynamics

B, T = int(os.environ.get("BATCH", 1024)), int(os.environ.get("T", 10))
dyn = DeviceDynamics("pendulum_dx")
n, m = 3, 1
rng = np.random.default_rng(0)
th = rng.uniform(-np.pi / 2, np.pi / 2, B)
x0 = torch.tensor(np.stack([np.cos(th), np.sin(th), rng.uniform(-1, 1, B)], 1), dtype=torch.float64, device="cuda")
goal = torch.tensor([1.0, 0.0, 0.0, 0.0], dtype=torch.float64, device="cuda")
Qw = torch.tensor([1.0, 1.0, 0.1, 0.001], dtype=torch.float64, device="cuda")
C = torch.diag(Qw).repeat(T, B, 1, 1).requires_grad_()
c = (-(Qw * goal)).repeat(T, B, 1).requires_grad_()
lo, hi = torch.tensor([-2.0], dtype=torch.float64, device="cuda"), torch.tensor([2.0], dtype=torch.float64, device="cuda")
for fused in (True, False):
    qp_wrapper.FUSED_MPC_QP = fused
    for tag, kw in (("single QP", dict(single_qp_solve=True)), ("SQP qp_iter=3", dict(qp_iter=3))):
        mpc = qp_wrapper.MPC(n, m, T, u_lower=lo, u_upper=hi, n_batch=B, verbose=-1, **kw)
        def step():
            x, u = mpc(x0, qp_wrapper.QuadCost(C, c), dyn, dyn.jac)
            (x.sum() + u.sum()).backward()
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 10
        for _ in range(reps): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        print("%-34s %-14s B=%d T=%d: %.2f ms per call (%.0f k trajectories/s)"
              % ("stage-wise kernels" if fused else "dense one-QP-per-wavefront kernels", tag, B, T, dt * 1e3, B / dt / 1e3))
qp_wrapper.FUSED_MPC_QP = True
