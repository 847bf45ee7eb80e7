#!/usr/bin/env python3
"""Golden vectors for the large-n MPC QP path (BASELINE config 4 state/control sizes): the reference's
qp_wrapper.MPC with LinDx dynamics at n_state 12, n_ctrl 4 ("goldens from the
reference's qp_wrapper.MPC at n=12, m=4, T<=6"), single-QP and SQP (qp_iter 3) modes, x, u and the
gradients wrt C, c, F, f, x0.  Build container only (imports the reference).

Problem family: per-sample SPD stage costs, per-sample linear dynamics x+ = (I + small) x + B u + f,
control bounds +-0.5 (a good share of the controls end on them).

Usage:  python tests/golden/make_golden_ric.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("DQP_REFERENCE", "/root/reference")
m_ = types.ModuleType("ipdb")
def _st(*a, **k):
    raise RuntimeError("ipdb.set_trace() reached inside the reference")
m_.set_trace = _st
sys.modules["ipdb"] = m_
sys.path.insert(0, REF)
torch.set_default_dtype(torch.float64)
from qpth import qp_wrapper  # noqa: E402


def family(seed, B, n, m, T):
    g = torch.Generator().manual_seed(seed)
    nt = n + m
    L = 0.3 * torch.randn(T, B, nt, nt, generator=g)
    C = L @ L.transpose(2, 3) + torch.eye(nt)
    c = torch.randn(T, B, nt, generator=g)
    F = torch.cat([torch.eye(n) + 0.1 * torch.randn(T - 1, B, n, n, generator=g),
                   0.5 * torch.randn(T - 1, B, n, m, generator=g)], dim=-1)
    f = 0.1 * torch.randn(T - 1, B, n, generator=g)
    x0 = torch.randn(B, n, generator=g)
    return dict(C=C, c=c, F=F, f=f, x0=x0, u_lower=-0.5 * torch.ones(m), u_upper=0.5 * torch.ones(m))


def run(name, B, n, m, T, seed):
    md = family(seed, B, n, m, T)
    outs = {"in_" + k: v.numpy() for k, v in md.items()}
    for tag, kw in (("single", dict(single_qp_solve=True)), ("sqp", dict(qp_iter=3))):
        ins = {k: md[k].clone().requires_grad_() for k in ("C", "c", "F", "f", "x0")}
        mpc = qp_wrapper.MPC(n, m, T, u_lower=md["u_lower"], u_upper=md["u_upper"], n_batch=B, verbose=-1, **kw)
        x, u = mpc(ins["x0"], qp_wrapper.QuadCost(ins["C"], ins["c"]), qp_wrapper.LinDx(ins["F"], ins["f"]), None)
        (x.sum() + 2.0 * u.sum()).backward()
        outs["%s_x" % tag] = x.detach().numpy()
        outs["%s_u" % tag] = u.detach().numpy()
        for k, t in ins.items():
            outs["%s_d%s" % (tag, k)] = t.grad.numpy() if t.grad is not None else np.zeros(t.shape)
        print(name, tag, "|u| max %.3f" % float(u.abs().max()), "share on a bound %.2f" % float((u.abs() > 0.4999).double().mean()))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **outs)


if __name__ == "__main__":
    run("RIC_n12_m4_T6_b4", 4, 12, 4, 6, seed=5)
    run("RIC_n12_m4_T10_b3", 3, 12, 4, 10, seed=6)
