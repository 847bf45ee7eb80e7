#!/usr/bin/env python3
"""Generate golden input/output vectors from the reference implementation.

Runs ONLY in the build container (needs /root/reference, which never travels to the
GPU box).  It imports the reference's Python package as-is (CPU, fp64), feeds it seeded
inputs of the families named in SURVEY.md §8(d) and stores inputs + outputs as small
.npz fixtures in this directory.  The fixtures are data only; no reference source is
copied.

The reference modules `import ipdb` (a debugger, not installed here) at module scope.
A stand-in module object is registered in sys.modules whose set_trace() raises, so a
hidden breakpoint on any exercised path fails loudly instead of hanging.

Reference entry points exercised:
  qpth/qp.py:19-183        QPFunction forward/backward   (block-LU PDIPM, batch.py)
  qpth/qp.py:187-271       DenseQPFunction forward/backward (full KKT, batch_LU.py)
  qpth/solvers/pdipm/batch.py:46-208,377-469   forward / pre_factor_kkt / factor_kkt
  qpth/qp_wrapper.py:638-679  compute_Qq_dense / compute_Ab_dense / compute_Gh_dense
  qpth/qp_wrapper.py:213-296  MPC.forward with LinDx dynamics

Usage:  python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("DQP_REFERENCE", "/root/reference")


def _install_ipdb_stub():
    m = types.ModuleType("ipdb")

    def set_trace(*a, **k):
        raise RuntimeError("ipdb.set_trace() reached inside the reference")

    m.set_trace = set_trace
    sys.modules["ipdb"] = m


_install_ipdb_stub()
sys.path.insert(0, REF)
torch.set_default_dtype(torch.float64)
torch.set_num_threads(8)

from qpth.qp import QPFunction, DenseQPFunction  # noqa: E402
from qpth.solvers.pdipm import batch as pdipm_b  # noqa: E402
from qpth import qp_wrapper  # noqa: E402


def bmv(M, x):
    return (M @ x.unsqueeze(-1)).squeeze(-1)


def family_R(seed, B, nz, nineq, neq, shared=()):
    """Random dense QP (SURVEY §8d family R; prof-linear.py:64-75 / test.py:42-55)."""
    g = torch.Generator().manual_seed(seed)
    L = torch.randn(B, nz, nz, generator=g)
    Q = L @ L.transpose(1, 2) + 1e-3 * torch.eye(nz)
    G = torch.randn(B, nineq, nz, generator=g)
    z0 = torch.randn(B, nz, generator=g)
    s0 = torch.rand(B, nineq, generator=g)
    A = torch.randn(B, neq, nz, generator=g) if neq > 0 else torch.zeros(B, 0, nz)
    p = torch.randn(B, nz, generator=g)
    d = dict(Q=Q, p=p, G=G, A=A)
    for k in shared:  # parameters shared across the batch (no batch dim)
        if k in d:
            d[k] = d[k][0].clone()
    Gb = d["G"] if d["G"].dim() == 3 else d["G"].unsqueeze(0).expand(B, -1, -1)
    Ab = d["A"] if d["A"].dim() == 3 else d["A"].unsqueeze(0).expand(B, -1, -1)
    d["h"] = bmv(Gb, z0) + s0
    d["b"] = bmv(Ab, z0) if neq > 0 else torch.zeros(B, 0)
    if "h" in shared:
        d["h"] = d["h"][0].clone()
    if "b" in shared:
        d["b"] = d["b"][0].clone()
    return d


def family_M(seed, B, n_state, n_ctrl, T):
    """MPC-structured QP (SURVEY §8d family M; examples/train.py:57-82), assembled by the
    reference's own compute_*_dense."""
    g = torch.Generator().manual_seed(seed)
    n_tau = n_state + n_ctrl
    A_dyn = torch.eye(n_state) + 0.2 * torch.randn(n_state, n_state, generator=g)
    B_dyn = torch.randn(n_state, n_ctrl, generator=g)
    C = torch.eye(n_tau).unsqueeze(0).unsqueeze(0).repeat(T, B, 1, 1)
    c = torch.randn(n_tau, generator=g).unsqueeze(0).unsqueeze(0).repeat(T, B, 1)
    x0 = torch.randn(B, n_state, generator=g)
    F = torch.cat([A_dyn, B_dyn], dim=1).unsqueeze(0).unsqueeze(0).repeat(T - 1, B, 1, 1)
    f = torch.zeros(T - 1, B, n_state)
    # 1-D bounds (n_ctrl,), as deqmpc/policies.py:581-582 passes them; compute_Gh_dense
    # (qp_wrapper.py:677-678) only broadcasts this shape.
    u_lower = -torch.ones(n_ctrl)
    u_upper = torch.ones(n_ctrl)
    return dict(C=C, c=c, F=F, f=f, x0=x0, u_lower=u_lower, u_upper=u_upper)


def run_qpfunction(d, ct_seed=123, maxIter=20):
    """QPFunction fwd (zhat, lam, nu, slack) + grads for cotangent=ones and a seeded one."""
    names = ["Q", "p", "G", "h", "A", "b"]
    out = {}
    neq = d["A"].shape[-2]
    grads_all = {}
    for tag in ("ones", "rand"):
        ins = [d[k].clone().requires_grad_() for k in names]
        Q_, p_, G_, h_, A_, b_ = ins
        B = max(t.shape[0] for t, nd in zip(ins, (3, 2, 3, 2, 3, 2)) if t.dim() == nd)

        def dyn_res(x):
            if neq == 0:
                return torch.zeros(x.shape[0], 0)
            Ae = A_ if A_.dim() == 3 else A_.unsqueeze(0).expand(B, -1, -1)
            return bmv(Ae, x) - b_

        def cost_grad(x):
            Qe = Q_ if Q_.dim() == 3 else Q_.unsqueeze(0).expand(B, -1, -1)
            return bmv(Qe, x) + p_

        # neq == 0: the fork's util.expandParam (util.py:68-75, the second definition) rejects
        # the 1-D empty torch.Tensor() the docstring advertises; a (B,0,nz)/(B,0) pair passes.
        A_in, b_in = A_, b_
        zhat = QPFunction(check_Q_spd=False, verbose=-1, maxIter=maxIter)(
            Q_, p_, G_, h_, A_in, b_in, dyn_res, cost_grad)
        if tag == "ones":
            ct = torch.ones_like(zhat)
        else:
            ct = torch.randn(zhat.shape, generator=torch.Generator().manual_seed(ct_seed))
        zhat.backward(ct)
        out["zhat"] = zhat.detach().numpy()
        out["ct_" + tag] = ct.numpy()
        for k, t in zip(names, ins):
            if neq == 0 and k in ("A", "b"):
                continue
            grads_all["d%s_%s" % (k, tag)] = t.grad.numpy()
    out.update(grads_all)

    # lam / nu / slack: what QPFunctionFn.forward stashes on ctx (qp.py:93-96).
    with torch.no_grad():
        B = out["zhat"].shape[0]

        def ex(t, nd):
            return t if t.dim() == nd else t.unsqueeze(0).expand(B, *t.shape)
        Q, p, G, h = ex(d["Q"], 3), ex(d["p"], 2), ex(d["G"], 3), ex(d["h"], 2)
        if neq > 0:
            A, b = ex(d["A"], 3), ex(d["b"], 2)
        else:
            A, b = torch.Tensor(), torch.Tensor()
        Q_LU, S_LU, R = pdipm_b.pre_factor_kkt(Q, G, A)
        zh, nus, lams, slacks = pdipm_b.forward(
            Q, p, G, h, A, b, Q_LU, S_LU, R,
            (lambda x: bmv(A, x) - b) if neq > 0 else (lambda x: torch.zeros(B, 0)),
            lambda x: bmv(Q, x) + p, 1e-12, -1, 3, maxIter)
        assert np.allclose(zh.numpy(), out["zhat"], rtol=0, atol=1e-12)
        out["lam"] = lams.numpy()
        out["slack"] = slacks.numpy()
        out["nu"] = nus.numpy() if neq > 0 else np.zeros((B, 0))
    return out


def run_dense(d, ct_seed=123):
    """DenseQPFunction fwd + grads (all params batched; qp.py:187-271)."""
    names = ["Q", "p", "G", "h", "A", "b"]
    out = {}
    for tag in ("ones", "rand"):
        ins = [d[k].clone().requires_grad_() for k in names]
        Q_, p_, G_, h_, A_, b_ = ins
        zhat = DenseQPFunction()(Q_, p_, G_, h_, A_, b_, lambda x: bmv(A_, x) - b_)
        if tag == "ones":
            ct = torch.ones_like(zhat)
        else:
            ct = torch.randn(zhat.shape, generator=torch.Generator().manual_seed(ct_seed))
        zhat.backward(ct)
        out["dense_zhat"] = zhat.detach().numpy()
        for k, t in zip(names, ins):
            out["dense_d%s_%s" % (k, tag)] = t.grad.numpy()
    return out


def save(name, d_in, outs):
    arrs = {"in_" + k: v.detach().numpy() for k, v in d_in.items()}
    arrs.update(outs)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %-28s %6.1f KB  zhat[0,:3]=%s" % (
        name + ".npz", os.path.getsize(path) / 1024, np.array2string(
            arrs.get("zhat", arrs.get("x", np.zeros((1, 3))))[0].ravel()[:3], precision=6)))


def main():
    # ---- family R ------------------------------------------------------------------
    cases = [
        ("R_metric_b8", dict(seed=0, B=8, nz=30, nineq=30, neq=15), True),
        ("R_small_b5", dict(seed=1, B=5, nz=10, nineq=5, neq=3), True),
        ("R_noeq_b4", dict(seed=2, B=4, nz=12, nineq=8, neq=0), False),
        ("R_cfg2shape_b4", dict(seed=3, B=4, nz=40, nineq=20, neq=30), True),
        ("R_wide_b3", dict(seed=4, B=3, nz=20, nineq=50, neq=4), True),
        ("R_one_b1", dict(seed=5, B=1, nz=6, nineq=4, neq=2), True),
    ]
    for name, kw, dense in cases:
        d = family_R(**kw)
        outs = run_qpfunction(d)
        if dense:
            outs.update(run_dense(d))
        save(name, d, outs)

    # parameters shared across the batch -> gradients are .mean(0) (qp.py:160-178)
    d = family_R(seed=6, B=6, nz=10, nineq=8, neq=4, shared=("Q", "G", "A"))
    save("R_shared_QGA_b6", d, run_qpfunction(d))
    d = family_R(seed=7, B=4, nz=8, nineq=6, neq=3, shared=("Q", "p", "G", "h", "A"))
    save("R_shared_all_but_b_b4", d, run_qpfunction(d))

    # ---- family M: assembled by the reference's compute_*_dense ----------------------
    for name, (B, n, m, T) in (("M_metric_b8", (8, 3, 3, 5)), ("M_pend_shape_b4", (4, 3, 1, 10))):
        md = family_M(42, B, n, m, T)
        mpc = qp_wrapper.MPC(n, m, T, u_lower=md["u_lower"], u_upper=md["u_upper"],
                             n_batch=B, verbose=-1)
        with torch.no_grad():
            Q, q = mpc.compute_Qq_dense(md["C"], md["c"])
            A, b = mpc.compute_Ab_dense(md["F"], md["f"], md["x0"])
            G, h = mpc.compute_Gh_dense(md["x0"])
        d = dict(Q=Q, p=q, G=G, h=h, A=A, b=b)
        outs = run_qpfunction(d)
        outs.update(run_dense(d))
        outs.update({"mpc_" + k: v.numpy() for k, v in md.items()})

        # Full qp_wrapper.MPC forward with LinDx (single QP + line search and qp_iter loop).
        for tag, kw in (("single", dict(single_qp_solve=True)), ("sqp", dict(qp_iter=3))):
            Cg = md["C"].clone().requires_grad_()
            cg = md["c"].clone().requires_grad_()
            Fg = md["F"].clone().requires_grad_()
            fg = md["f"].clone().requires_grad_()
            x0g = md["x0"].clone().requires_grad_()
            mpc = qp_wrapper.MPC(n, m, T, u_lower=md["u_lower"], u_upper=md["u_upper"],
                                 n_batch=B, verbose=-1, **kw)
            x, u = mpc(x0g, qp_wrapper.QuadCost(Cg, cg), qp_wrapper.LinDx(Fg, fg), None)
            (x.sum() + 2.0 * u.sum()).backward()
            outs["mpc_%s_x" % tag] = x.detach().numpy()
            outs["mpc_%s_u" % tag] = u.detach().numpy()
            for k, t in (("C", Cg), ("c", cg), ("F", Fg), ("f", fg), ("x0", x0g)):
                outs["mpc_%s_d%s" % (tag, k)] = (
                    t.grad.numpy() if t.grad is not None else np.zeros(t.shape))
        save(name, d, outs)


if __name__ == "__main__":
    main()
