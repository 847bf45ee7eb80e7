#!/usr/bin/env python3
"""Kernel-level timing of the fused MPC QP (dqp_mpc_qp_forward / _backward) against the dense
pipeline (dqp_mpc_assemble + dqp_qp_forward / dqp_qp_backward + dqp_mpc_assemble_backward) at the
metric shape n=3 m=3 T=5, B=4096: HIP events around each C-ABI call."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diff_qp_mpc_amd import _lib, qp_wrapper
lib = _lib.load()
n, m, T = 3, 3, 5
B = int(os.environ.get("BATCH", "4096"))
gen = torch.Generator().manual_seed(42)
Ad = torch.eye(n, dtype=torch.float64) + 0.2 * torch.randn(n, n, generator=gen, dtype=torch.float64)
Bd = torch.randn(n, m, generator=gen, dtype=torch.float64)
C = torch.eye(n + m, dtype=torch.float64).repeat(T, B, 1, 1).cuda()
c = torch.randn(T, B, n + m, generator=gen, dtype=torch.float64).cuda()
x0 = torch.randn(B, n, generator=gen, dtype=torch.float64).cuda()
F = torch.cat([Ad, Bd], 1).repeat(T - 1, B, 1, 1).cuda()
f = torch.zeros(T - 1, B, n, dtype=torch.float64).cuda()
one = torch.ones(m, dtype=torch.float64).cuda()


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


from diff_qp_mpc_amd import qp as qpmod
for term in ("batch", "per_problem"):
    qpmod.TERMINATION = term
    with torch.no_grad():
        t_f = timed(lambda: qp_wrapper._MPCQP.apply(C, c, F, f, x0, -one, one, n, m, T))
        def dense():
            Q, q, G, h, A, b = qp_wrapper._AssembleDenseQP.apply(C, c, F, f, x0, -one, one, n, m, T)
            return qpmod.DenseQPFunction()(Q, q, G, h, A, b, None)
        t_d = timed(dense)
    print("termination %-12s fused MPC QP forward %.3f ms   assemble + dense forward %.3f ms" % (term, t_f, t_d))
