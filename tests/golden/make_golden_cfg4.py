#!/usr/bin/env python3
"""Golden vectors for BASELINE config 4: AL_mpc.MPC on the quadrotor of deqmpc/rex_quadrotor.py
(n_state 12, n_ctrl 4), produced by running the reference (build container only).

The reference's quadrotor is a torch module (RexQuadrotor_dynamics, RK4 on a rigid body with MRP
attitude) with an autograd Jacobian class (RexQuadrotor_dynamics_jac); both are imported as they
are (without the torch.jit.script wrapper RexQuadrotor adds -- same arithmetic).  Modules are built
under the default float32 dtype like the reference's scripts do (their constants are float32
tensors), the solver runs in float64 (train.py --dtype double).
Solver setup as deqmpc/policies.py:567-639 (Tracking_MPC, solver_type "al"), weights and bounds of
the env (rex_quadrotor.py:167-172: Q = [10,10,10, .01x3, 1x3, .01x3], R = 1e-4, 11.5 <= u <= 18.3),
initial states inside the env's reset window (attitude as MRP directly), tracking reference a
straight line from x0 to the origin with the hover command (in DEQ-MPC it comes from the network).
Cases: the full config-4 horizon T = 30 (nz = 480, B = 4) and T = 6 (B = 4); two successive forward
calls each (cold start, then the history warm start), gradients of the first.

Usage:  python tests/golden/make_golden_cfg4.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("DQP_REFERENCE", "/root/reference")
m = types.ModuleType("ipdb")
def _st(*a, **k):
    raise RuntimeError("ipdb.set_trace() reached inside the reference")
m.set_trace = _st
sys.modules["ipdb"] = m
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "deqmpc"))

from rex_quadrotor import RexQuadrotor_dynamics, RexQuadrotor_dynamics_jac  # noqa: E402
dyn, dyn_jac = RexQuadrotor_dynamics(), RexQuadrotor_dynamics_jac()              # float32 constants
torch.set_default_dtype(torch.float64)
dyn_jac.identity = dyn_jac.identity.double()
from qpth import AL_mpc, al_utils  # noqa: E402


def run_case(name, B, T, seed):
    nx, nu, dt = 12, 4, dyn.dt
    rng = np.random.default_rng(seed)
    win = np.array([1.0] * 3 + [0.15] * 3 + [0.5] * 3 + [0.25] * 3)
    x0 = torch.tensor(rng.uniform(-1, 1, (B, nx)) * win)
    u_upper, u_lower = torch.tensor([18.3] * nu), torch.tensor([11.5] * nu)
    Qw = torch.tensor([10.0] * 3 + [0.01] * 3 + [1.0] * 3 + [0.01] * 3 + [1e-4] * nu)
    Qd = Qw.repeat(B, T, 1)
    ramp = torch.linspace(1.0, 0.0, T)[None, :, None]
    x_ref = x0[:, None, :] * ramp
    hover = (2.0 * 9.81 - 4 * dyn.bf) / (4 * dyn.kf * dyn.act_scale)
    u_ref = torch.full((B, T, nu), float(hover))
    xu_ref = torch.cat([x_ref, u_ref], dim=-1)
    C = torch.diag_embed(Qd).requires_grad_()
    c = (-(Qd * xu_ref)).clone().requires_grad_()
    ctrl = AL_mpc.MPC(nx, nu, T, u_lower=u_lower, u_upper=u_upper, n_batch=B, verbose=0,
                      u_init=torch.randn(B, T, nu), solver_type="dense", dtype=torch.float64, eps=1e-5,
                      exit_unconverged=False, backprop=False)
    ctrl.reinitialize(x0, torch.ones(B, T, 1))
    ctrl.x_init, ctrl.u_init = x_ref.clone(), u_ref.clone()           # policies.py:644-646
    out = {}
    x, u = ctrl(x0, al_utils.QuadCost(C, c), dyn, dyn_jac)
    (x.double().sum() + 2.0 * u.double().sum()).backward()
    out.update(x1=x.detach().numpy(), u1=u.detach().numpy(),
               lam1=ctrl.lamda_prev.detach().numpy(), rho1=ctrl.rho_prev.detach().numpy(),
               dC1=C.grad.diagonal(dim1=-2, dim2=-1).numpy().copy(), dc1=c.grad.numpy().copy())
    C2, c2 = C.detach().clone().requires_grad_(), c.detach().clone().requires_grad_()
    x2, u2 = ctrl(x0, al_utils.QuadCost(C2, c2), dyn, dyn_jac)
    out.update(x2=x2.detach().numpy(), u2=u2.detach().numpy(),
               lam2=ctrl.lamda_prev.detach().numpy(), rho2=ctrl.rho_prev.detach().numpy())
    arrs = dict(in_x0=x0.numpy(), in_Qd=Qd.numpy(), in_c=c.detach().numpy(), in_u_lower=u_lower.numpy(),
                in_u_upper=u_upper.numpy(), in_x_init=x_ref.numpy(), in_u_init=u_ref.numpy(), dt=dt)
    arrs.update(out)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrs)
    xs, us = x.detach().double(), u.detach().double()
    gap = (dyn(xs[:, :-1].reshape(-1, nx), us[:, :-1].reshape(-1, nu)).reshape(B, T - 1, nx) - xs[:, 1:]).abs().max()
    print("wrote", name, "u range %.3f..%.3f" % (float(us.min()), float(us.max())),
          "dynamics gap of the returned trajectory %.2e" % float(gap), "rho", ctrl.rho_prev.reshape(-1)[:3].numpy())


if __name__ == "__main__":
    torch.manual_seed(0)
    run_case("CFG4_rexquadrotor_T30_b4", B=4, T=30, seed=0)
    run_case("CFG4_rexquadrotor_T6_b4", B=4, T=6, seed=1)
