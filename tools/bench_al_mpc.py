#!/usr/bin/env python3
"""End-to-end timing of the AL_mpc.MPC mirror (rows a14-a17) on the pendulum of deqmpc/envs.py
(n=2, m=1), B=4096, T=20: one forward (2 AL iterations x 4 Newton steps, each = Jacobian
assembly + dqp_al_newton_step + 20-way line search) + backward, with a cProfile of the host side."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from diff_qp_mpc_amd import AL_mpc, al_utils

class Pendulum(torch.nn.Module):
    dt, g, m, l = 0.05, 10.0, 1.0, 1.0
    def forward(self, x, u):
        th, thd = x[..., 0], x[..., 1]
        acc = (u.squeeze(-1) + self.m * self.g * self.l * torch.sin(th)) / (self.m * self.l ** 2)
        nthd = thd + acc * self.dt
        return torch.stack((th + nthd * self.dt, nthd), dim=-1)

class PendulumJac(Pendulum):
    def forward(self, x, u):
        xn = Pendulum.forward(self, x, u)
        c = self.g * torch.cos(x[..., 0]) / self.l
        N = x.shape[0]
        fx = x.new_zeros(N, 2, 2)
        fx[:, 1, 0] = c * self.dt; fx[:, 1, 1] = 1.0; fx[:, 0, 0] = 1.0 + c * self.dt ** 2; fx[:, 0, 1] = self.dt
        fu = x.new_zeros(N, 2, 1)
        fu[:, 1, 0] = self.dt; fu[:, 0, 0] = self.dt ** 2
        return xn, (fx, fu)

B, T, nx, nu = int(os.environ.get("BATCH", 4096)), int(os.environ.get("T", 20)), 2, 1
gen = torch.Generator().manual_seed(0)
x0 = (torch.rand(B, nx, generator=gen, dtype=torch.float64) * 2 - 1).cuda()
Qd = torch.ones(B, T, nx + nu, dtype=torch.float64).cuda(); Qd[..., nx:] = 1e-2
C = torch.diag_embed(Qd).requires_grad_()
c = torch.zeros(B, T, nx + nu, dtype=torch.float64).cuda().requires_grad_()
lim = torch.full((nu,), 2.0, dtype=torch.float64).cuda()
ctrl = AL_mpc.MPC(nx, nu, T, u_lower=-lim, u_upper=lim, n_batch=B, verbose=0, solver_type="dense",
                  dtype=torch.float64, eps=1e-5, exit_unconverged=False, backprop=False)
dyn, dyn_jac = Pendulum(), PendulumJac()
if os.environ.get("DEVICE_DYN", "1") == "1":       # registered device model: fused NewtonAL path
    from diff_qp_mpc_amd.dynamics import DeviceDynamics
    dyn = DeviceDynamics("pendulum_euler")
    dyn_jac = dyn.jac
    AL_mpc.FUSED_NEWTON_AL = os.environ.get("FUSED", "1") == "1"
def step():
    ctrl.reinitialize(x0, torch.ones(B, T, 1, device="cuda"))
    x, u = ctrl(x0, al_utils.QuadCost(C, c), dyn, dyn_jac)
    (x.double().sum() + 2.0 * u.double().sum()).backward()
for _ in range(2):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
print("AL_mpc.MPC pendulum B=%d T=%d: forward+backward %.2f ms per call (%.1f k trajectories/s)" % (B, T, dt * 1e3, B / dt / 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
