#!/usr/bin/env python3
"""End-to-end timing of AL_mpc.MPC on the registered robots at the BASELINE shapes:
  ROBOT=cartpole1l   BATCH=4096 T=20   (config 3)
  ROBOT=rexquadrotor BATCH=8192 T=30   (config 4: n 12, m 4, nz 480)
  ROBOT=cartpole2l   BATCH=8192 T=5    (config 5 robot, per-GPU share of the 65536 batch)
One call = 2 AL iterations x 4 Newton steps (block-tridiagonal factorisation + 20-candidate line
search each) + the backward solve; setup as the golden generators (tests/golden/make_golden_cfg3.py,
make_golden_cfg4.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from diff_qp_mpc_amd import AL_mpc, al_utils
from diff_qp_mpc_amd.dynamics import DeviceDynamics

robot = os.environ.get("ROBOT", "rexquadrotor")
B = int(os.environ.get("BATCH", 8192 if robot != "cartpole1l" else 4096))
T = int(os.environ.get("T", {"cartpole1l": 20, "rexquadrotor": 30, "cartpole2l": 5}.get(robot, 20)))
dyn = DeviceDynamics(robot)
nx, nu = dyn.n_state, dyn.n_ctrl
rng = np.random.default_rng(0)
f64 = dict(dtype=torch.float64, device="cuda")
if robot == "rexquadrotor":
    win = np.array([1.0] * 3 + [0.15] * 3 + [0.5] * 3 + [0.25] * 3)
    x0 = torch.tensor(rng.uniform(-1, 1, (B, nx)) * win, **f64)
    Qw = torch.tensor([10.0] * 3 + [0.01] * 3 + [1.0] * 3 + [0.01] * 3 + [1e-4] * nu, **f64)
    lo, hi = torch.full((nu,), 11.5, **f64), torch.full((nu,), 18.3, **f64)
    u_ref = torch.full((B, T, nu), (2.0 * 9.81 + 4 * 30.48576) / (4 * 0.0244101 * 100.0), **f64)
else:
    x0 = torch.tensor(rng.uniform(-np.pi, np.pi, (B, nx)), **f64)
    Qw = torch.cat([torch.ones(nx), 1e-8 * torch.ones(nu)]).to(**f64)
    ub = 100.0 if robot == "cartpole1l" else 250.0
    lo, hi = torch.full((nu,), -ub, **f64), torch.full((nu,), ub, **f64)
    u_ref = torch.zeros(B, T, nu, **f64)
Qd = Qw.repeat(B, T, 1)
x_ref = x0[:, None, :] * torch.linspace(1.0, 0.0, T, **f64)[None, :, None]
C = torch.diag_embed(Qd).requires_grad_()
c = (-(Qd * torch.cat([x_ref, u_ref], -1))).clone().requires_grad_()
ctrl = AL_mpc.MPC(nx, nu, T, u_lower=lo, u_upper=hi, n_batch=B, verbose=0, solver_type="dense",
                  dtype=torch.float64, eps=1e-5, exit_unconverged=False, backprop=False)
mask = torch.ones(B, T, 1, device="cuda")

def step(backward=True):
    ctrl.reinitialize(x0, mask)
    ctrl.x_init, ctrl.u_init = x_ref, u_ref
    x, u = ctrl(x0, al_utils.QuadCost(C, c), dyn, dyn.jac)
    if backward:
        (x.double().sum() + 2.0 * u.double().sum()).backward()
    return x, u

for _ in range(2):
    x, u = step()
assert bool(torch.isfinite(x).all())
reps = int(os.environ.get("REPS", 5))
res = {}
for name, bw in (("forward", False), ("forward+backward", True)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        step(bw)
    torch.cuda.synchronize(); res[name] = (time.perf_counter() - t0) / reps
print("AL_mpc.MPC %s B=%d T=%d (nz=%d): forward %.2f ms, forward+backward %.2f ms per call (%.1f k trajectories/s)"
      % (robot, B, T, T * (nx + nu), res["forward"] * 1e3, res["forward+backward"] * 1e3, B / res["forward+backward"] / 1e3))
xs, us = x.double(), u.double()
gap = (dyn(xs[:, :-1].reshape(-1, nx), us[:, :-1].reshape(-1, nu)).reshape(B, T - 1, nx) - xs[:, 1:]).abs().amax(dim=(1, 2))
print("dynamics gap of the returned trajectories: median %.2e max %.2e; rho %s" % (float(gap.median()), float(gap.max()), ctrl.rho_prev.unique().tolist()))
