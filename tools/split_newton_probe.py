#!/usr/bin/env python3
"""Would the Newton step be faster as two kernels -- the model's Jacobians for all knots in parallel (jac_kernel), then the
block-tridiagonal sweep on them (al_banded_newton_kernel<Given>) -- than as the fused kernel?  Kernel times of both paths
from the library's own tracing, at config-5 (cartpole-2, B 65536, T 5) and config-3 / config-4 sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff_qp_mpc_amd import AL_mpc, al_utils, _lib
from diff_qp_mpc_amd.dynamics import DeviceDynamics


class Opaque(torch.nn.Module):
    def __init__(self, d):
        super().__init__(); self.d = d
    def forward(self, x, u):
        return self.d(x, u)


for robot, B, T in (("cartpole2l", 65536, 5), ("cartpole1l", 4096, 20), ("rexquadrotor", 8192, 30)):
    d = DeviceDynamics(robot)
    nx, nu = d.n_state, d.n_ctrl
    gen = torch.Generator().manual_seed(0)
    x0 = ((torch.rand(B, nx, generator=gen, dtype=torch.float64) * 2 - 1) * 0.3).cuda()
    Qd = torch.ones(B, T, nx + nu, dtype=torch.float64).cuda(); Qd[..., nx:] = 1e-2
    C = torch.diag_embed(Qd); c = torch.zeros(B, T, nx + nu, dtype=torch.float64).cuda()
    lim = torch.full((nu,), 20.0, dtype=torch.float64).cuda()
    for name, dyn in (("fused", d), ("split", Opaque(d))):
        ctrl = AL_mpc.MPC(nx, nu, T, u_lower=-lim, u_upper=lim, n_batch=B, verbose=0, solver_type="dense",
                          dtype=torch.float64, eps=1e-5, exit_unconverged=False, backprop=False)
        def step():
            ctrl.reinitialize(x0, torch.ones(B, T, 1, device="cuda"))
            return ctrl(x0, al_utils.QuadCost(C, c), dyn, d.jac)
        for _ in range(2): step()
        torch.cuda.synchronize()
        with _lib.trace(4096) as tr:
            step(); torch.cuda.synchronize()
        print(robot, B, T, name)
        for k, (cnt, ms) in sorted(tr.by_kernel().items(), key=lambda kv: -kv[1][0] * kv[1][1]):
            print("    %-90s x%3d  %8.3f ms total  %8.1f us avg" % (k[:90], cnt, ms * cnt, ms * 1e3))
