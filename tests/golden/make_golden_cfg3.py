#!/usr/bin/env python3
"""Golden vectors for BASELINE config 3: AL_mpc.MPC on the 1-link cartpole (n_state 4, n_ctrl 1,
T = 20), produced by running the reference (build container only).

The reference's cartpole dynamics is a compiled extension (deqmpc/my_envs/cartpole.py imports
cartpole1l, a CUDAExtension this image cannot build).  Its two layers are used separately here:
  * deqmpc/my_envs/dynamics.py `Dynamics` (the Python wrapper: state <-> (q, qdot, tau, h),
    Jacobian block concatenation / transposes) is imported as is;
  * its `package` (dynamics(q, qdot, tau, h), derivatives(...)) is the reference's own
    CasADi-generated C compiled by oracle/Makefile into oracle/_ref and called through
    oracle/dyn_ref.py -- the same expressions the extension's CPU path evaluates
    (cartpole1l/src/dynamics_cpu.cpp:8-27).
Solver setup as deqmpc/policies.py:567-639 (Tracking_MPC, solver_type "al") with the cartpole
env's weights (deqmpc/my_envs/cartpole.py:66-78: Q = 1, R = 1e-8, |u| <= 100, dt = 0.05) and
initial states as CartpoleEnv.reset (uniform +-pi on q and qdot).  The tracking reference is a
straight line from x0 to the upright origin (in DEQ-MPC it comes from the network).
Two successive forward calls are stored (cold start, then the history warm start).

Usage:  python tests/golden/make_golden_cfg3.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("DQP_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
m = types.ModuleType("ipdb")
def _st(*a, **k):
    raise RuntimeError("ipdb.set_trace() reached inside the reference")
m.set_trace = _st
sys.modules["ipdb"] = m
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "deqmpc", "my_envs"))
torch.set_default_dtype(torch.float64)

from qpth import AL_mpc, al_utils  # noqa: E402
from dynamics import Dynamics  # noqa: E402   (deqmpc/my_envs/dynamics.py)
from oracle import dyn_ref  # noqa: E402


class RefPackage:
    """What `import cartpole1l` provides in the reference: dynamics / derivatives on torch tensors."""

    def __init__(self, robot):
        self.robot = robot

    def dynamics(self, q, qdot, tau, h):
        a, b = dyn_ref.dynamics(self.robot, q.numpy(), qdot.numpy(), tau.numpy(), h.numpy().reshape(-1))
        return torch.tensor(a), torch.tensor(b)

    def derivatives(self, q, qdot, tau, h):
        return tuple(torch.tensor(b) for b in
                     dyn_ref.derivatives(self.robot, q.numpy(), qdot.numpy(), tau.numpy(), h.numpy().reshape(-1)))


def run_case(name, robot, B, T, seed):
    nq = dyn_ref.ROBOTS[robot]
    nx, nu, dt = 2 * nq, 1, 0.05
    dyn = Dynamics(nx=nx, dt=dt, kwargs=dict(dtype=torch.float64))
    dyn.package = RefPackage(robot)
    rng = np.random.default_rng(seed)
    x0 = torch.tensor(rng.uniform(-np.pi, np.pi, (B, nx)))
    ub = 100.0 if nq == 2 else 250.0
    u_upper, u_lower = torch.tensor([ub]), torch.tensor([-ub])
    Qd = torch.cat([torch.ones(nx), 1e-8 * torch.ones(nu)]).repeat(B, T, 1)
    ramp = torch.linspace(1.0, 0.0, T)[None, :, None]
    x_ref = x0[:, None, :] * ramp
    u_ref = torch.zeros(B, T, nu)
    xu_ref = torch.cat([x_ref, u_ref], dim=-1)
    C = torch.diag_embed(Qd).requires_grad_()
    c = (-(Qd * xu_ref)).clone().requires_grad_()
    ctrl = AL_mpc.MPC(nx, nu, T, u_lower=u_lower, u_upper=u_upper, n_batch=B, verbose=0,
                      u_init=torch.randn(B, T, nu), solver_type="dense", dtype=torch.float64, eps=1e-5,
                      exit_unconverged=False, backprop=False)
    ctrl.reinitialize(x0, torch.ones(B, T, 1))
    ctrl.x_init, ctrl.u_init = x_ref.clone(), u_ref.clone()           # policies.py:644-646
    out = {}
    x, u = ctrl(x0, al_utils.QuadCost(C, c), dyn, dyn.dynamics_derivatives)
    (x.double().sum() + 2.0 * u.double().sum()).backward()
    out.update(x1=x.detach().numpy(), u1=u.detach().numpy(),
               lam1=ctrl.lamda_prev.detach().numpy(), rho1=ctrl.rho_prev.detach().numpy(),
               dC1=C.grad.diagonal(dim1=-2, dim2=-1).numpy().copy(), dc1=c.grad.numpy().copy())
    C2, c2 = C.detach().clone().requires_grad_(), c.detach().clone().requires_grad_()
    x2, u2 = ctrl(x0, al_utils.QuadCost(C2, c2), dyn, dyn.dynamics_derivatives)
    out.update(x2=x2.detach().numpy(), u2=u2.detach().numpy(),
               lam2=ctrl.lamda_prev.detach().numpy(), rho2=ctrl.rho_prev.detach().numpy())
    arrs = dict(in_x0=x0.numpy(), in_Qd=Qd.numpy(), in_c=c.detach().numpy(), in_u_lower=u_lower.numpy(),
                in_u_upper=u_upper.numpy(), in_x_init=x_ref.numpy(), in_u_init=u_ref.numpy(), dt=dt)
    arrs.update(out)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrs)
    xs, us = x.detach().double(), u.detach().double()
    gap = (dyn(xs[:, :-1].reshape(-1, nx), us[:, :-1].reshape(-1, nu)).reshape(B, T - 1, nx) - xs[:, 1:]).abs().max()
    print("wrote", name, "max|u| %.3f" % float(us.abs().max()), "dynamics gap of the returned trajectory %.2e" % float(gap),
          "rho", ctrl.rho_prev.reshape(-1)[:3].numpy())


if __name__ == "__main__":
    torch.manual_seed(0)
    run_case("CFG3_cartpole1l_T20_b4", "cartpole1l", B=4, T=20, seed=0)
    run_case("CFG5_cartpole2l_T5_b4", "cartpole2l", B=4, T=5, seed=1)
