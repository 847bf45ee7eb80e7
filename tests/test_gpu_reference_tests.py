"""The reference's own gradient tests (test.py:42-187: test_dl_dp / dG / dh / dA / db), restated on the
MI355X operators: same problem generator (npr.seed(1), nz 10, the same scale factors), same loss
1/2 ||zhat - truez||^2, same tolerances (RTOL 1e-4, ATOL 1e-2).  The reference differentiates its cvxpy
forward numerically (numdifftools); cvxpy is not installed here, so the numerical derivative is a central
difference of THIS package's forward -- the analytic gradient of the KKT backward pass against the
numerical one of the solver it belongs to."""
import numpy as np
import numpy.random as npr
import pytest
import torch

pytestmark = pytest.mark.gpu
ATOL, RTOL = 1e-2, 1e-4


def problem(nz=10, neq=1, nineq=3, Qscale=1., Gscale=1., Ascale=1.):
    npr.seed(1)
    L = np.random.randn(nz, nz)
    Q = Qscale * L.dot(L.T)
    G = Gscale * npr.randn(nineq, nz)
    z0 = npr.randn(nz)
    s0 = npr.rand(nineq)
    h = G.dot(z0) + s0
    A = Ascale * npr.randn(neq, nz)
    b = A.dot(z0)
    p = npr.randn(1, nz)
    truez = npr.randn(1, nz)
    return dict(Q=Q, p=p[0], G=G, h=h, A=A, b=b), truez


def solve(d, grad=False):
    import diff_qp_mpc_amd as dqp
    t = {k: torch.tensor(v, dtype=torch.float64, device="cuda").requires_grad_(grad) for k, v in d.items()}
    A, b = (t["A"], t["b"]) if d["A"].shape[0] > 0 else (torch.empty(0, device="cuda", dtype=torch.float64),) * 2
    z = dqp.QPFunction(verbose=-1)(t["Q"], t["p"].unsqueeze(0), t["G"], t["h"], A, b)
    return z, t


def loss_of(d, truez):
    z, _ = solve(d)
    return 0.5 * float(((z.cpu().numpy() - truez) ** 2).sum())


def analytic(d, truez):
    z, t = solve(d, grad=True)
    z.backward(z.detach() - torch.tensor(truez, dtype=torch.float64, device="cuda"))
    return {k: (v.grad.cpu().numpy() if v.grad is not None else None) for k, v in t.items()}


def numeric(d, truez, key, eps=1e-6):
    base = d[key]
    out = np.zeros_like(base)
    it = np.nditer(base, flags=["multi_index"])
    for _ in it:
        i = it.multi_index
        hi, lo = dict(d), dict(d)
        hi[key] = base.copy(); hi[key][i] += eps
        lo[key] = base.copy(); lo[key][i] -= eps
        out[i] = (loss_of(hi, truez) - loss_of(lo, truez)) / (2 * eps)
    return out


def test_dl_dp():
    d, truez = problem(nz=10, neq=2, nineq=3, Qscale=100., Gscale=100., Ascale=100.)
    np.testing.assert_allclose(numeric(d, truez, "p"), analytic(d, truez)["p"], rtol=RTOL, atol=ATOL)


def test_dl_dG():
    d, truez = problem(nz=10, neq=0, nineq=3)
    np.testing.assert_allclose(numeric(d, truez, "G"), analytic(d, truez)["G"], rtol=RTOL, atol=ATOL)


def test_dl_dh():
    d, truez = problem(nz=10, neq=0, nineq=3, Qscale=1., Gscale=1.)
    np.testing.assert_allclose(numeric(d, truez, "h"), analytic(d, truez)["h"], rtol=RTOL, atol=ATOL)


def test_dl_dA():
    d, truez = problem(nz=10, neq=3, nineq=1, Qscale=100., Gscale=100., Ascale=100.)
    np.testing.assert_allclose(numeric(d, truez, "A"), analytic(d, truez)["A"], rtol=RTOL, atol=ATOL)


def test_dl_db():
    d, truez = problem(nz=10, neq=3, nineq=1, Qscale=100., Gscale=100., Ascale=100.)
    np.testing.assert_allclose(numeric(d, truez, "b"), analytic(d, truez)["b"], rtol=RTOL, atol=ATOL)
