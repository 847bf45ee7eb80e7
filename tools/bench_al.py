#!/usr/bin/env python3
"""Timing of dqp_al_newton_step at the BASELINE config-3 shape (cartpole-1, T=20: nz=100,
ncon = T n + 2 T m = 120, B=4096) with a block-banded clamped Jacobian like the real one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diff_qp_mpc_amd import al_utils

B, n, m, T = 4096, 4, 1, 20
nt, nz, neq, nineq = n + m, T * (n + m), T * n, 2 * T * m
g = torch.Generator(device="cuda").manual_seed(0)
J = torch.zeros(B, neq + nineq, nz, dtype=torch.float64, device="cuda")
Jeq = J[:, :neq].view(B, T, n, T, nt)
ar = torch.arange(T - 1, device="cuda")
Jeq[:, ar, :, ar, :] = -torch.randn(T - 1, B, n, nt, dtype=torch.float64, device="cuda", generator=g)
Jeq[:, ar, :, ar + 1, :n] = torch.eye(n, dtype=torch.float64, device="cuda")
Jeq[:, T - 1, :, 0, :n] = torch.eye(n, dtype=torch.float64, device="cuda")
Jiq = J[:, neq:].view(B, T, 2, m, T, nt)
at = torch.arange(T, device="cuda")
act = (torch.rand(B, T, 2, m, device="cuda", generator=g) > 0.7).double()
Jiq[:, at, 0, :, at, n:] = (torch.eye(m, dtype=torch.float64, device="cuda") * act[:, :, 0, :, None]).permute(1, 0, 2, 3)
Jiq[:, at, 1, :, at, n:] = (-torch.eye(m, dtype=torch.float64, device="cuda") * act[:, :, 1, :, None]).permute(1, 0, 2, 3)
Qd = torch.rand(B, nz, dtype=torch.float64, device="cuda", generator=g) + 0.01
rho = torch.full((B, 1), 10.0, dtype=torch.float64, device="cuda")
grad = torch.randn(B, nz, dtype=torch.float64, device="cuda", generator=g)
terms = al_utils.HessianTerms(J, Qd, rho)
H = terms.dense()
U0, _ = torch.linalg.cholesky_ex(H)
ref0 = -torch.cholesky_solve(grad[:, :, None], U0)[:, :, 0]     # reference taken once, up front
torch.cuda.synchronize()
for _ in range(3):
    upd, L, info = al_utils.newton_step(terms, grad)
torch.cuda.synchronize()
reps = int(os.environ.get("REPS", "20"))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    upd, L, info = al_utils.newton_step(terms, grad)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
# torch reference of the same step on the same GPU (rocBLAS bmm + rocSOLVER potrf/potrs)
for _ in range(2):
    U, inf2 = torch.linalg.cholesky_ex(terms.dense()); ref = -torch.cholesky_solve(grad[:, :, None], U)[:, :, 0]
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    U, inf2 = torch.linalg.cholesky_ex(terms.dense()); ref = -torch.cholesky_solve(grad[:, :, None], U)[:, :, 0]
torch.cuda.synchronize(); tms = (time.perf_counter() - t0) / 5 * 1e3
alg_bytes = B * ((neq + nineq) * nz + 2 * nz + 1 + nz + nz * nz) * 8
flops = B * (2.0 * nz * nz * (neq + nineq) / 2 + nz ** 3 / 3 + 2 * nz * nz)
print("dqp_al_newton_step B=%d nz=%d ncon=%d: %.3f ms/step  (%.1f k problems/s)  %.0f GB/s algorithmic (%.1f%% of 8 TB/s), %.2f TFLOP/s fp64"
      % (B, nz, neq + nineq, ms, B / ms, alg_bytes / ms / 1e6, alg_bytes / ms / 1e6 / 80, flops / ms / 1e9))
print("same step with torch ops on this GPU (bmm + cholesky_ex + cholesky_solve): %.3f ms  -> speed-up %.1fx" % (tms, tms / ms))
res = lambda v: float((torch.bmm(H, v[:, :, None])[:, :, 0] + grad).abs().max())
print("max |update - torch| = %.2e, max |L - torch| = %.2e, info max %d, residual |H u + g|: kernel %.2e torch %.2e"
      % (float((upd - ref0).abs().max()), float((L - U0).abs().max()), int(info.max()), res(upd), res(ref0)))
print("(torch result of the timed back-to-back loop: residual %.2e)" % res(ref))
