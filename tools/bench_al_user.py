#!/usr/bin/env python3
"""AL_mpc.MPC with a caller-supplied dynamics MODULE (torch nn.Module with its own Jacobian function, what the reference's
envs are): the dense Newton step (dqp_al_assemble + dqp_al_newton_step: Hessian by fp64 MFMA, cyclic LDL^T in registers;
nz <= 128) against the block-tridiagonal step on the module's Jacobians (dqp_al_banded_newton_step_jac) at the same
sizes -- one call = 2 AL iterations x 4 Newton steps + backward, pendulum of deqmpc/envs.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from diff_qp_mpc_amd import AL_mpc, al_utils
from test_gpu_al import Pendulum, PendulumJac

nx, nu = 2, 1
for B, T in ((4096, 20), (4096, 40), (128, 20)):
    gen = torch.Generator().manual_seed(0)
    x0 = (torch.rand(B, nx, generator=gen, dtype=torch.float64) * 2 - 1).cuda()
    Qd = torch.ones(B, T, nx + nu, dtype=torch.float64).cuda(); Qd[..., nx:] = 1e-2
    C = torch.diag_embed(Qd).requires_grad_()
    c = torch.zeros(B, T, nx + nu, dtype=torch.float64).cuda().requires_grad_()
    lim = torch.full((nu,), 2.0, dtype=torch.float64).cuda()
    dyn, dyn_jac = Pendulum(), PendulumJac()
    out = {}
    for name, thr in (("dense Newton step", 128), ("block-tridiagonal on the module's Jacobians", 0)):
        AL_mpc.BANDED_USER_DYNAMICS_FROM_NZ = thr
        ctrl = AL_mpc.MPC(nx, nu, T, u_lower=-lim, u_upper=lim, n_batch=B, verbose=0, solver_type="dense",
                          dtype=torch.float64, eps=1e-5, exit_unconverged=False, backprop=False)
        def step():
            ctrl.reinitialize(x0, torch.ones(B, T, 1, device="cuda"))
            x, u = ctrl(x0, al_utils.QuadCost(C, c), dyn, dyn_jac)
            (x.double().sum() + 2.0 * u.double().sum()).backward()
            return x, u
        for _ in range(2): x, u = step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        out[name] = (dt, x.detach().clone())
        print("B=%d T=%d nz=%d  %-46s %.2f ms per call" % (B, T, T * (nx + nu), name, dt * 1e3))
    a, b = [v[1] for v in out.values()]
    print("   max |x| difference between the two paths: %.2e" % float((a - b).abs().max()))
