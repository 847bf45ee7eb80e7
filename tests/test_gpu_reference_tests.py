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


@pytest.mark.parametrize("family_flag", ["auto", "rows", "generic"])
def test_kkt_solver_against_a_dense_solve(family_flag):
    """The reference's test_lu_kkt_solver / test_ir_kkt_solver (test.py:190-247) compare its block KKT solve
    with a dense LU of the full KKT matrix on a random problem.  Here: the one KKT solve of the backward
    kernels (every family) at an arbitrary interior point (s, z > 0, not a solution) against numpy's dense
    solve of  [Q 0 G' A'; 0 D I 0; G I 0 0; A 0 0 0] [dx ds dz dy] = -[g 0 0 0],  D = z/s  -- the gradients
    the kernels return are dp = dx, dh = -dz, db = -dy (qp.py:143-178)."""
    from diff_qp_mpc_amd import qp as qpmod, _lib
    rng = np.random.default_rng(0)
    B, nz, nineq, neq = 5, 30, 30, 15
    L = rng.standard_normal((B, nz, nz))
    Q = L @ L.transpose(0, 2, 1) + 1e-3 * np.eye(nz)
    G = rng.standard_normal((B, nineq, nz)); A = rng.standard_normal((B, neq, nz))
    s = rng.random((B, nineq)) + 0.1; z = rng.random((B, nineq)) + 0.1
    zhat = rng.standard_normal((B, nz)); nu = rng.standard_normal((B, neq)); g = rng.standard_normal((B, nz))
    dp_ref = np.zeros((B, nz)); dh_ref = np.zeros((B, nineq)); db_ref = np.zeros((B, neq))
    for i in range(B):
        D = np.diag(z[i] / s[i])
        Z = np.zeros
        K = np.block([[Q[i], Z((nz, nineq)), G[i].T, A[i].T],
                      [Z((nineq, nz)), D, np.eye(nineq), Z((nineq, neq))],
                      [G[i], np.eye(nineq), Z((nineq, nineq)), Z((nineq, neq))],
                      [A[i], Z((neq, nineq)), Z((neq, nineq)), Z((neq, neq))]])
        sol = np.linalg.solve(K, -np.concatenate([g[i], np.zeros(2 * nineq + neq)]))
        dp_ref[i] = sol[:nz]; dh_ref[i] = -sol[nz + nineq:nz + 2 * nineq]; db_ref[i] = -sol[nz + 2 * nineq:]
    t = lambda a: torch.tensor(a, dtype=torch.float64, device="cuda")
    old = qpmod.FORCE_FLAGS
    qpmod.FORCE_FLAGS = {"auto": 0, "rows": _lib.DQP_FLAG_NO_NULLSPACE, "generic": _lib.DQP_FLAG_GENERIC_ONLY}[family_flag]
    try:
        dims = _lib.dqp_dims(B, nz, nineq, neq, nz * nz, nz, nineq * nz, nineq, neq * nz, neq)
        out = qpmod._backward_impl((t(Q), t(G), t(A), dims, None), t(zhat), t(z), t(nu), t(s), t(g), (True,) * 6,
                                   _lib.DQP_FLAG_DENSE_BACKWARD)
    finally:
        qpmod.FORCE_FLAGS = old
    dQ, dp, dG, dh, dA, db = [o.cpu().numpy() for o in out]
    np.testing.assert_allclose(dp, dp_ref, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(dh, dh_ref, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(db, db_ref, rtol=1e-8, atol=1e-10)
