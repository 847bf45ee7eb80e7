#!/usr/bin/env python3
"""Golden vectors for the AL_mpc / NewtonAL row, produced by importing
the reference (build container only; see make_golden.py for the ipdb stand-in).

Reference entry points exercised:
  qpth/AL_mpc.py:116-321      MPC.__init__/forward/al_solve, reinitialize (:432-438)
  qpth/al_utils.py:363-500    NewtonAL forward/backward (4 Newton steps, Cholesky, line search)
  qpth/al_utils.py:37-102,162-318  merit_function, merit_grad_hessian, constraint_res_jac2, ...
  deqmpc/envs.py:5-82         PendulumDynamics / PendulumDynamics_jac (semi-implicit Euler)

Stored per case: inputs (x0, Q diag as full C, p, u bounds, x_init, u_init), outputs of two
successive MPC.forward calls (the second one exercises the warm start, al_utils.py:16-34), the
solver state (lamda_prev, rho_prev), the gradients wrt C's diagonal and c of a linear loss, and
one isolated Newton step (grad, J_clamp, Qdiag, rho -> Hessian, Cholesky solve) for the kernel
that replaces it.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("DQP_REFERENCE", "/root/reference")
m = types.ModuleType("ipdb")
def _st(*a, **k):
    raise RuntimeError("ipdb.set_trace() reached inside the reference")
m.set_trace = _st
sys.modules["ipdb"] = m
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "deqmpc"))
torch.set_default_dtype(torch.float64)

from qpth import AL_mpc, al_utils  # noqa: E402
import envs  # noqa: E402  (deqmpc/envs.py)


def run_case(name, B, T, seed, lin=False):
    g = torch.Generator().manual_seed(seed)
    env = envs.PendulumEnv(stabilization=False)
    nx, nu = env.nx, env.nu
    dyn, dyn_jac = env.dynamics, env.dynamics_derivatives
    u_upper = torch.tensor(env.action_space.high, dtype=torch.float64)
    u_lower = torch.tensor(env.action_space.low, dtype=torch.float64)
    x0 = torch.stack([3.0 * (torch.rand(B, generator=g) - 0.5), torch.rand(B, generator=g) - 0.5], 1)
    Qd = torch.cat([torch.tensor([10.0, 1.0]), torch.tensor([0.01])]).repeat(B, T, 1)
    x_ref = torch.zeros(B, T, nx + nu)
    x_ref[:, :, 0] = 0.3 * torch.randn(B, T, generator=g)
    C = torch.diag_embed(Qd).requires_grad_()
    c = (-(Qd * x_ref)).clone().requires_grad_()
    u_init = 0.1 * torch.randn(B, T, nu, generator=g)
    x_init = None

    ctrl = AL_mpc.MPC(nx, nu, T, u_lower=u_lower, u_upper=u_upper, n_batch=B, verbose=0,
                      u_init=u_init, solver_type="dense", dtype=torch.float64, eps=1e-5,
                      exit_unconverged=False, backprop=False)
    ctrl.reinitialize(x0, torch.ones(B, T, 1))
    ctrl.u_init = u_init
    out = {}
    x, u = ctrl(x0, al_utils.QuadCost(C, c), dyn, dyn_jac)
    (x.double().sum() + 2.0 * u.double().sum()).backward()
    out.update(x1=x.detach().numpy(), u1=u.detach().numpy(),
               lam1=ctrl.lamda_prev.detach().numpy(), rho1=ctrl.rho_prev.detach().numpy(),
               dC1=C.grad.diagonal(dim1=-2, dim2=-1).numpy().copy(), dc1=c.grad.numpy().copy())
    # second call: warm start from history + previous x_init/u_init
    C2 = C.detach().clone().requires_grad_()
    c2 = c.detach().clone().requires_grad_()
    x2, u2 = ctrl(x0, al_utils.QuadCost(C2, c2), dyn, dyn_jac)
    out.update(x2=x2.detach().numpy(), u2=u2.detach().numpy(),
               lam2=ctrl.lamda_prev.detach().numpy(), rho2=ctrl.rho_prev.detach().numpy())

    # one isolated Newton step at the first iterate (what the HIP kernel computes)
    with torch.no_grad():
        xu = torch.cat((ctrl.rollout(x0, u_init, dyn), u_init), dim=2)
    lam = torch.zeros(B, nx * T + 2 * nu * T)
    rho = torch.ones(B, 1)
    xu_ = xu.clone().requires_grad_(True)
    grad, Hess = al_utils.merit_grad_hessian(xu_, Qd, c.detach(), dyn, dyn_jac, x0, lam, rho,
                                             None, None, u_lower, u_upper, True)
    res, res_clamp, J, Jc, chess = al_utils.constraint_res_jac2(xu_, x0, dyn_jac, None, None, u_lower, u_upper)
    U, info = torch.linalg.cholesky_ex(Hess)
    upd = -torch.cholesky_solve(grad.reshape(B, -1, 1), U).reshape(B, -1)
    out.update(ns_grad=grad.detach().numpy(), ns_Jc=Jc.detach().numpy(), ns_J=J.detach().numpy(),
               ns_res=res.detach().numpy(), ns_res_clamp=res_clamp.detach().numpy(),
               ns_H=Hess.detach().numpy(), ns_L=U.detach().numpy(), ns_update=upd.detach().numpy(),
               ns_xu=xu.numpy(), ns_rho=rho.numpy(), ns_Qd=Qd.numpy())
    ins = dict(x0=x0.numpy(), Qd=Qd.numpy(), c=c.detach().numpy(), u_lower=u_lower.numpy(),
               u_upper=u_upper.numpy(), u_init=u_init.numpy())
    arrs = {"in_" + k: v for k, v in ins.items()}
    arrs.update(out)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %s %.1f KB  x1[0,:2]=%s u1[0,:2]=%s rho1=%s" % (
        name, os.path.getsize(path) / 1024, x.detach().numpy()[0, :2].ravel(),
        u.detach().numpy()[0, :2].ravel(), ctrl.rho_prev.detach().numpy()[0]))


if __name__ == "__main__":
    run_case("AL_pendulum_T5_b8", B=8, T=5, seed=0)
    run_case("AL_pendulum_T10_b6", B=6, T=10, seed=1)
