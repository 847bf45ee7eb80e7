#!/usr/bin/env python3
"""Golden for qp_wrapper.MPC(add_goal_constraint=True) (qp_wrapper.py:338-345,638-656): run in the
build container against the reference itself (same ipdb accommodation as make_golden.py).
Output: tests/golden/G_goal_b4.npz (inputs, x, u, gradients of x.sum() + 2 u.sum())."""
import os, sys, types
import numpy as np
import torch

REF = os.environ.get("DQP_REFERENCE", "/root/reference")
stub = types.ModuleType("ipdb")
def _boom(*a, **k):
    raise RuntimeError("ipdb.set_trace() reached in the reference")
stub.set_trace = _boom
sys.modules["ipdb"] = stub
sys.path.insert(0, REF)
from qpth import qp_wrapper  # noqa: E402

torch.set_default_dtype(torch.float64)
B, n, m, T = 4, 3, 2, 4
gen = torch.Generator().manual_seed(11)
Ad = torch.eye(n) + 0.2 * torch.randn(n, n, generator=gen)
Bd = torch.randn(n, m, generator=gen)
C = torch.eye(n + m).repeat(T, B, 1, 1) * (0.5 + torch.rand(T, B, 1, 1, generator=gen))
c = torch.randn(T, B, n + m, generator=gen)
x0 = 0.3 * torch.randn(B, n, generator=gen)
F = torch.cat([Ad, Bd], 1).repeat(T - 1, B, 1, 1)
f = torch.zeros(T - 1, B, n)
ul, uu = -3.0 * torch.ones(m), 3.0 * torch.ones(m)
ins = dict(C=C, c=c, F=F, f=f, x0=x0)
g = {k: v.clone().requires_grad_() for k, v in ins.items()}
mpc = qp_wrapper.MPC(n, m, T, u_lower=ul, u_upper=uu, n_batch=B, verbose=-1, single_qp_solve=True,
                     add_goal_constraint=True, x_goal=torch.zeros(B, n))
x, u = mpc(g["x0"], qp_wrapper.QuadCost(g["C"], g["c"]), qp_wrapper.LinDx(g["F"], g["f"]), None)
(x.sum() + 2.0 * u.sum()).backward()
out = {"in_" + k: v.numpy() for k, v in ins.items()}
out.update(in_u_lower=ul.numpy(), in_u_upper=uu.numpy(), x=x.detach().numpy(), u=u.detach().numpy())
for k, t in g.items():
    out["d" + k] = t.grad.numpy() if t.grad is not None else np.zeros(t.shape)
print("x_T (goal = 0):", np.abs(out["x"][-1]).max())
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "G_goal_b4.npz"), **out)
print("saved G_goal_b4.npz")
