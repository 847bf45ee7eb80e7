#!/usr/bin/env python3
"""Timing of the stage-wise MPC QP kernels (csrc/dqp_ric.hip) at the BASELINE config-4 shape:
n_state 12, n_ctrl 4, T 30 (nz 480, nineq 240, neq 360), B = 8192 (BATCH), HIP events around the
C-ABI calls dqp_mpc_qp_forward / dqp_mpc_qp_backward, both termination modes."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diff_qp_mpc_amd import _lib
lib = _lib.load()
n, m, T = int(os.environ.get("N", 12)), int(os.environ.get("M", 4)), int(os.environ.get("T", 30))
B = int(os.environ.get("BATCH", 8192))
nt = n + m
gen = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float64, device="cuda")
L = 0.3 * rnd(T, B, nt, nt)
C = L @ L.transpose(2, 3) + torch.eye(nt, dtype=torch.float64, device="cuda")
del L
c = rnd(T, B, nt)
F = torch.cat([torch.eye(n, dtype=torch.float64, device="cuda") + 0.05 * rnd(T - 1, B, n, n), 0.5 * rnd(T - 1, B, n, m)], -1).contiguous()
f = 0.1 * rnd(T - 1, B, n)
x0 = rnd(B, n)
lo, hi = torch.full((m,), -0.4, dtype=torch.float64, device="cuda"), torch.full((m,), 0.4, dtype=torch.float64, device="cuda")
dims = _lib.dqp_mpc_dims(B, n, m, T, 1, 0)
assert lib.dqp_mpc_qp_supported(ctypes.byref(dims)) == 1
kw = dict(dtype=torch.float64, device="cuda")
tau = torch.empty(B, T, nt, **kw); lam = torch.empty(B, 2 * T * m, **kw); slack = torch.empty(B, 2 * T * m, **kw)
nu = torch.empty(B, T * n, **kw); info = torch.empty(B, 2, dtype=torch.int32, device="cuda"); resid = torch.empty(B, **kw)
wsb = int(lib.dqp_mpc_qp_workspace_bytes(ctypes.byref(dims)))
ws = torch.empty(wsb // 8, **kw)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
g = torch.ones(B, T, nt, **kw)
dC, dc, dF, df, dx0 = torch.empty_like(C), torch.empty_like(c), torch.empty_like(F), torch.empty_like(f), torch.empty_like(x0)
print("B=%d n=%d m=%d T=%d  nz=%d  workspace %.0f MB  inputs %.0f MB" % (B, n, m, T, T * nt, wsb / 1e6, (C.numel() + F.numel()) * 8 / 1e6))


def timed(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2]


for mode, flag in (("batch", _lib.DQP_FLAG_BATCH_TERMINATION), ("per_problem", 0)):
    opts = _lib.dqp_opts(1e-12, 1e-10, 20, 3, flag, 0)
    tb = int(lib.dqp_mpc_qp_termination_bytes(ctypes.byref(dims), ctypes.byref(opts)))
    term = torch.empty(max(tb // 8, 1), **kw)
    def fwd():
        rc = lib.dqp_mpc_qp_forward(ctypes.byref(dims), ctypes.byref(opts), P(C), P(c), P(F), P(f), P(x0), P(lo), P(hi),
                                    P(tau), P(lam), P(nu), P(slack), P(info), P(resid), P(ws), P(term) if tb else None, None)
        assert rc == 0, rc
    tf = timed(fwd)
    it = info[:, 1].float()
    print("termination %-12s forward %.2f ms (%.0f k QP/s)   iterations mean %.1f max %d   best residual max %.1e   status!=0: %d"
          % (mode, tf, B / tf, float(it.mean()), int(it.max()), float(resid.max()), int((info[:, 0] != 0).sum())))
bo = _lib.dqp_opts(0.0, 0.0, 0, 0, _lib.DQP_FLAG_DENSE_BACKWARD, 0)
def bwd():
    rc = lib.dqp_mpc_qp_backward(ctypes.byref(dims), ctypes.byref(bo), P(C), P(F), P(tau), P(lam), P(nu), P(slack), P(g),
                                 P(dC), P(dc), P(dF), P(df), P(dx0), None, P(ws), None)
    assert rc == 0, rc
tb_ = timed(bwd)
print("backward %.2f ms" % tb_)
