#!/usr/bin/env python3
"""Golden vectors for BASELINE config 2: qp_wrapper.MPC on the nonlinear PendulumDx
(qpth/env_dx/pendulum.py:18-83; n_state 3, n_ctrl 1, T = 10, control bounds +-2), produced by
importing the reference (build container only).

What the reference does on this path and the fixture pins: the PDIPM inside every QP evaluates the
TRUE-dynamics residual closure (qp_wrapper.py:309,316 -> batch_LU.py:97), the QP is assembled from
the linearisation along the current trajectory (linearize_dynamics with dx_jac), the line search
rolls out the true dynamics.  Inputs as the reference's IL_Env builds them
(examples/il_env_nonconvex.py:57-65,81-104): x0 = (cos th, sin th, thdot), th ~ U(-pi/2, pi/2),
thdot ~ U(-1, 1), seed 0; cost from PendulumDx.get_true_obj (C = diag(q), c = p).
dx_jac: autograd of the module, in the form of the reference's own *_jac classes
(deqmpc/envs.py:68-82).  Cases: single_qp_solve and the SQP loop (qp_iter = 3), each with
gradients of sum(x) + 2 sum(u) wrt C and c.

Usage:  python tests/golden/make_golden_cfg2.py
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
torch.set_default_dtype(torch.float64)
torch.set_num_threads(8)

from qpth import qp_wrapper  # noqa: E402
from qpth.env_dx.pendulum import PendulumDx  # noqa: E402

B, T = 6, 10
dx = PendulumDx()
n, mm = dx.n_state, dx.n_ctrl


def dx_jac(x, u):
    """(x_next, (df/dx, df/du)) by autograd, one output row at a time."""
    with torch.enable_grad():
        xr, ur = x.detach().clone().requires_grad_(), u.detach().clone().requires_grad_()
        xn = dx(xr, ur)
        Jx, Ju = [], []
        for i in range(n):
            gx, gu = torch.autograd.grad(xn[:, i].sum(), [xr, ur], retain_graph=True)
            Jx.append(gx); Ju.append(gu)
    return xn.detach(), (torch.stack(Jx, 1), torch.stack(Ju, 1))


torch.manual_seed(0)
th = torch.rand(B) * np.pi - np.pi / 2
thdot = torch.rand(B) * 2 - 1
x0 = torch.stack((torch.cos(th), torch.sin(th), thdot), dim=1)
q, p = dx.get_true_obj()
out = dict(x0=x0.numpy(), q=q.numpy(), p=p.numpy(), dt=dx.dt, u_lower=dx.lower, u_upper=dx.upper)

for tag, kw in (("single", dict(single_qp_solve=True)), ("sqp3", dict(qp_iter=3))):
    C = torch.diag(q)[None, None].repeat(T, B, 1, 1).requires_grad_()
    c = p[None, None].repeat(T, B, 1).requires_grad_()
    ctrl = qp_wrapper.MPC(n, mm, T, u_lower=torch.tensor([dx.lower]), u_upper=torch.tensor([dx.upper]),
                          n_batch=B, max_linesearch_iter=dx.max_linesearch_iter,
                          linesearch_decay=dx.linesearch_decay, **kw)
    # the step factor of the LAST line search (the differentiable step is scaled by it, qp_wrapper.py:405-413): at a
    # converged iterate it is decided by round-off of the cost comparison, so it is recorded next to the gradients
    alphas = []
    ls = ctrl.line_search
    def recording_line_search(*a, **k):
        r = ls(*a, **k)
        alphas.append(r[2].detach().reshape(-1).numpy().copy())
        return r
    ctrl.line_search = recording_line_search
    x, u = ctrl(x0, qp_wrapper.QuadCost(C, c), dx, dx_jac)
    (x.sum() + 2.0 * u.sum()).backward()
    out[tag + "_alpha"] = alphas[-1]
    out[tag + "_x"], out[tag + "_u"] = x.detach().numpy(), u.detach().numpy()
    out[tag + "_dC"], out[tag + "_dc"] = C.grad.numpy(), c.grad.numpy()
    # dynamics consistency of the reference's own answer (line search rolls the true model out)
    xs = x.detach()
    gap = (dx(xs[:-1].reshape(-1, n), u.detach()[:-1].reshape(-1, mm)).reshape(T - 1, B, n) - xs[1:]).abs().max()
    print(tag, "x", tuple(x.shape), "max |u|", float(u.abs().max()), "rollout gap of the returned (x,u):", float(gap))

np.savez_compressed(os.path.join(HERE, "CFG2_pendulumdx_T10_b6.npz"), **out)
