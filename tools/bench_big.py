#!/usr/bin/env python3
"""The reference profiler's sizes (prof-linear.py:38-46: nz = nineq in {10, 50, 100, 500}, neq = 0, nBatch 128) and the
l1-slack MPC shape (90, 90, 15) through dqp_qp_forward + dqp_qp_backward: ms per solve, per-kernel times from the
library trace, and the MFMA-tile flop rate of the blocked kernels (csrc/dqp_big.hip)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from diff_qp_mpc_amd import qp as qpmod, _lib
from families import family

B = int(os.environ.get("BATCH", 128))
for nz, nineq, neq in ((10, 10, 0), (50, 50, 0), (100, 100, 0), (500, 500, 0), (90, 90, 15)):
    ins = [torch.tensor(a, device="cuda") for a in family(1, B, nz, nineq, neq, "R")]
    ct = torch.ones(B, nz, dtype=torch.float64, device="cuda")
    def step():
        z, l, n, s, info, r, saved = qpmod._forward_impl(*ins, 1e-12, 20, 3)
        qpmod._backward_impl(saved, z, l, n, s, ct, (True,) * 6, 0)
        return info, r
    for _ in range(2): info, r = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 5 if nz >= 500 else 20
    for _ in range(reps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    with _lib.trace(64) as tr:
        step(); torch.cuda.synchronize()
    ks = ", ".join("%s %.3f ms" % (k.split("(")[0].replace("void ", "").replace("dqp::", "")[:48], ms) for k, ms in tr.records)
    its = float(info[:, 1].float().mean())
    NP = (max(nz, 1) + 63) // 64 * 64
    flops = B * (its + 1) * NP ** 3 / 3.0 + B * (NP ** 3 / 3.0 + 1.5 * NP ** 3)      # potrf(T) per iteration + setup (potrf Q, trsm, syrk)
    print("nz=nineq=%d neq=%d B=%d: %.2f ms fwd+bwd (%.1f k QP/s), iterations %.1f, converged %.3f | %s%s"
          % (nz, neq, B, dt * 1e3, B / dt / 1e3, its, float((r < 1e-7).float().mean()), ks,
             (" | blocked O(n^3) flops %.2f TFLOP/s" % (flops / dt / 1e12)) if nz > 64 else ""))
