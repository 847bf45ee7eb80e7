#!/usr/bin/env python3
"""Golden vectors for the device dynamics registry (include/dqp.h dqp_dyn_*).

Robots: the reference's CasADi-generated C (deqmpc/my_envs/{pendulum1l,cartpole1l,cartpole2l}/src/
generated_dynamics.c, generated_derivatives.c) compiled where it lies into oracle/_ref by
oracle/Makefile and called through oracle/dyn_ref.py with the convention of
cartpole1l/src/dynamics_cpu.cpp:8-27.  Pendulum modules: the reference's torch classes imported
from /root/reference (deqmpc/envs.py PendulumDynamics / PendulumDynamics_jac,
qpth/env_dx/pendulum.py PendulumDx with Jacobians from torch.autograd).
Build container only; the fixtures are data (inputs + outputs).

Usage:  python tests/golden/make_golden_dyn.py
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
sys.path.insert(0, os.path.join(REF, "deqmpc"))
torch.set_default_dtype(torch.float64)

from oracle import dyn_ref  # noqa: E402

N = 48
for robot, nq in dyn_ref.ROBOTS.items():
    rng = np.random.default_rng(100 + nq)
    q = rng.uniform(-np.pi, np.pi, (N, nq)); qd = rng.uniform(-5, 5, (N, nq))
    tau = rng.uniform(-30, 30, (N, nq)); h = rng.uniform(0.01, 0.08, N)
    q[0] = 0; qd[0] = 0; tau[0] = 0                      # the equilibrium
    qo, qdo = dyn_ref.dynamics(robot, q, qd, tau, h)
    blocks = dyn_ref.derivatives(robot, q, qd, tau, h)
    x = np.concatenate([q, qd], 1); u = tau[:, :1].copy()
    xn = dyn_ref.step_x(robot, x, u, 0.05)
    Jx, Ju = dyn_ref.jac_x(robot, x, u, 0.05)
    np.savez_compressed(os.path.join(HERE, "DYN_%s.npz" % robot), q=q, qd=qd, tau=tau, h=h, q_out=qo,
                        qd_out=qdo, **{"blk%d" % i: b for i, b in enumerate(blocks)},
                        x=x, u=u, dt=0.05, x_next=xn, Jx=Jx, Ju=Ju)
    print(robot, "ok", np.abs(xn).max())

# deqmpc/envs.py pendulum (semi-implicit Euler) and its autograd Jacobian class
import envs as ref_envs  # noqa: E402
dyn, jac = ref_envs.PendulumDynamics(), ref_envs.PendulumDynamics_jac()
jac.identity = jac.identity.cpu()
rng = np.random.default_rng(7)
x = torch.tensor(rng.uniform(-4, 4, (N, 2))); u = torch.tensor(rng.uniform(-3, 3, (N, 1)))
xn = dyn(x, u)
xj, (Jx, Ju) = jac(x.clone().requires_grad_(), u.clone().requires_grad_())
np.savez_compressed(os.path.join(HERE, "DYN_pendulum_euler.npz"), x=x.numpy(), u=u.numpy(), dt=dyn.dt,
                    x_next=xn.numpy(), Jx=Jx.detach().numpy(), Ju=Ju.detach().numpy())
print("pendulum_euler ok", float((xj - xn).abs().max()))

# qpth/env_dx/pendulum.py PendulumDx (config 2): Jacobians by autograd, one output at a time
from qpth.env_dx.pendulum import PendulumDx  # noqa: E402
pdx = PendulumDx()
th = rng.uniform(-np.pi, np.pi, N)
x = torch.tensor(np.stack([np.cos(th), np.sin(th), rng.uniform(-6, 6, N)], 1))
u = torch.tensor(rng.uniform(-3, 3, (N, 1)))             # some beyond the +-2 clamp
u[1, 0] = 2.0                                            # exactly at the clamp bound
xr, ur = x.clone().requires_grad_(), u.clone().requires_grad_()
xn = pdx(xr, ur)
Jx = torch.zeros(N, 3, 3); Ju = torch.zeros(N, 3, 1)
for i in range(3):
    gx, gu = torch.autograd.grad(xn[:, i].sum(), [xr, ur], retain_graph=True)
    Jx[:, i] = gx; Ju[:, i] = gu
np.savez_compressed(os.path.join(HERE, "DYN_pendulum_dx.npz"), x=x.numpy(), u=u.numpy(), dt=pdx.dt,
                    x_next=xn.detach().numpy(), Jx=Jx.numpy(), Ju=Ju.numpy())
print("pendulum_dx ok")

# deqmpc/rex_quadrotor.py (BASELINE config 4): the torch module and its autograd Jacobian class, built
# under the default float32 dtype like the reference's scripts (its constants are float32 tensors),
# evaluated on float64 states
torch.set_default_dtype(torch.float32)
from rex_quadrotor import RexQuadrotor_dynamics, RexQuadrotor_dynamics_jac  # noqa: E402
qd_, qj_ = RexQuadrotor_dynamics(), RexQuadrotor_dynamics_jac()
torch.set_default_dtype(torch.float64)
rng = np.random.default_rng(11)
win = np.array([5.0] * 3 + [0.4] * 3 + [0.5] * 3 + [0.25] * 3)
x = torch.tensor(rng.uniform(-1, 1, (N, 12)) * win); u = torch.tensor(rng.uniform(11.5, 18.3, (N, 4)))
xn = qd_(x, u)
xj, (Jx, Ju) = qj_(x.clone().requires_grad_(), u.clone().requires_grad_())
np.savez_compressed(os.path.join(HERE, "DYN_rexquadrotor.npz"), x=x.numpy(), u=u.numpy(), dt=qd_.dt,
                    x_next=xn.numpy(), Jx=Jx.detach().numpy(), Ju=Ju.detach().numpy())
print("rexquadrotor ok", float((xj - xn).abs().max()), float(xn.abs().max()))
