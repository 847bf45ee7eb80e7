"""GPU parity of the AL_mpc / NewtonAL row.

  * dqp_al_newton_step / dqp_al_chol_solve (HIP, fp64 MFMA Hessian + LDS Cholesky) against the
    reference's own Newton-step data (tests/golden/AL_*.npz) and the numpy oracle at larger sizes:
    L rtol 1e-9, update rtol 1e-8;
  * AL_mpc.MPC end to end (two successive forward calls: cold start, then the history warm start)
    with the pendulum dynamics of deqmpc/envs.py against the reference's outputs: x, u (float32 in
    the reference, AL_mpc.py:319-320) rtol 1e-4 / atol 1e-5; multipliers rtol 1e-5; gradients of a
    linear loss wrt the cost diagonal and c rtol 1e-4 / atol 1e-6.
"""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import al_oracle

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
AL_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "AL_*.npz")))


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def dev(a, grad=False):
    t = torch.tensor(np.asarray(a), dtype=torch.float64, device="cuda")
    return t.requires_grad_() if grad else t


class Pendulum(torch.nn.Module):
    """deqmpc/envs.py:5-47 PendulumDynamics (semi-implicit Euler), restated."""
    dt, g, m, l = 0.05, 10.0, 1.0, 1.0

    def forward(self, x, u):
        th, thd = x[..., 0], x[..., 1]
        acc = (u.squeeze(-1) + self.m * self.g * self.l * torch.sin(th)) / (self.m * self.l ** 2)
        nthd = thd + acc * self.dt
        return torch.stack((th + nthd * self.dt, nthd), dim=-1)


class PendulumJac(Pendulum):
    """Same map as envs.py:62-76 (there via autograd): returns x_next and (df/dx, df/du)."""

    def forward(self, x, u):
        xn = Pendulum.forward(self, x, u)
        c = self.g * torch.cos(x[..., 0]) / self.l
        N = x.shape[0]
        fx = x.new_zeros(N, 2, 2)
        fx[:, 1, 0] = c * self.dt
        fx[:, 1, 1] = 1.0
        fx[:, 0, 0] = 1.0 + c * self.dt ** 2
        fx[:, 0, 1] = self.dt
        fu = x.new_zeros(N, 2, 1)
        fu[:, 1, 0] = self.dt
        fu[:, 0, 0] = self.dt ** 2
        return xn, (fx, fu)


@pytest.mark.parametrize("name", AL_CASES)
def test_newton_step_kernel_vs_reference(name):
    from diff_qp_mpc_amd import al_utils
    g = load(name)
    B = g["ns_grad"].shape[0]
    terms = al_utils.HessianTerms(dev(g["ns_Jc"]), dev(g["ns_Qd"].reshape(B, -1)), dev(g["ns_rho"]))
    upd, L, info = al_utils.newton_step(terms, dev(g["ns_grad"]))
    assert int(info.abs().max()) == 0
    np.testing.assert_allclose(L.cpu().numpy(), g["ns_L"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(upd.cpu().numpy(), g["ns_update"], rtol=1e-8, atol=1e-11)
    rhs = np.random.default_rng(0).standard_normal(g["ns_grad"].shape)
    out = al_utils.chol_solve_neg(L, dev(rhs))
    np.testing.assert_allclose(out.cpu().numpy(), al_oracle.chol_solve_neg(g["ns_L"], rhs), rtol=1e-8, atol=1e-11)


@pytest.mark.parametrize("B,nz,ncon", [(37, 100, 120), (5, 128, 40), (300, 15, 30), (9, 1, 3), (4, 33, 0)])
def test_newton_step_kernel_vs_oracle(B, nz, ncon):
    """cartpole-T20 shape (100 x 120), the 128 limit, tiny and constraint-free problems."""
    from diff_qp_mpc_amd import al_utils
    rng = np.random.default_rng(nz)
    Jc = rng.standard_normal((B, ncon, nz)) * (rng.random((B, ncon, 1)) > 0.3)
    Qd = rng.random((B, nz)) + 0.05
    rho = 10.0 ** rng.integers(0, 3, (B, 1)).astype(np.float64)
    grad = rng.standard_normal((B, nz))
    upd, L, info = al_utils.newton_step(al_utils.HessianTerms(dev(Jc), dev(Qd), dev(rho)), dev(grad))
    ou, oL, oi = al_oracle.newton_update(Jc, Qd, rho, grad)
    assert int(info.abs().max()) == 0 and not oi.any()
    np.testing.assert_allclose(L.cpu().numpy(), oL, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(upd.cpu().numpy(), ou, rtol=1e-7, atol=1e-10)


def test_newton_step_reports_indefinite_hessian():
    from diff_qp_mpc_amd import al_utils
    B, nz = 3, 6
    Jc = np.zeros((B, 2, nz)); Qd = np.ones((B, nz)); Qd[1, 3] = -1.0
    upd, L, info = al_utils.newton_step(al_utils.HessianTerms(dev(Jc), dev(Qd), dev(np.ones((B, 1)))),
                                        dev(np.ones((B, nz))))
    assert info.cpu().tolist() == [0, 4, 0]                      # cholesky_ex: first bad leading minor
    u = upd.cpu().numpy()
    assert np.isnan(u[1]).all() and np.allclose(u[0], -1.0) and np.allclose(u[2], -1.0)


@pytest.mark.parametrize("B,T", [(5, 4), (33, 20), (1, 2)])
def test_assemble_jacobian_kernel(B, T):
    """dqp_al_assemble vs the index-scatter construction of al_utils.py:162-318 (restated in torch
    in al_utils.constraint_jacobian and in numpy in oracle/al_oracle.py) and the two bmm's of
    merit_grad_hessian: exact for Jc (copies and +-1), 1e-12 for the products."""
    from diff_qp_mpc_amd import al_utils
    gen = torch.Generator().manual_seed(T)
    n, m = 2, 1
    xu = torch.randn(B, T, n + m, generator=gen, dtype=torch.float64).cuda()
    x0 = torch.randn(B, n, generator=gen, dtype=torch.float64).cuda()
    lam = torch.randn(B, T * n + 2 * T * m, generator=gen, dtype=torch.float64).cuda()
    rho = (10.0 ** torch.randint(0, 3, (B, 1), generator=gen).double()).cuda()
    lo, hi = torch.full((m,), -0.5, dtype=torch.float64).cuda(), torch.full((m,), 0.5, dtype=torch.float64).cuda()
    res, resc, J, Jc_ref = al_utils.constraint_jacobian(xu, x0, PendulumJac(), lo, hi)
    g_ref = torch.bmm(lam[:, None], J)[:, 0] + rho * torch.bmm(resc[:, None], Jc_ref)[:, 0]
    Jc, gterm = al_utils.assemble_jacobian(xu, x0, PendulumJac(), lam, rho, lo, hi)
    assert torch.equal(Jc, Jc_ref)
    np.testing.assert_allclose(gterm.cpu().numpy(), g_ref.cpu().numpy(), rtol=1e-12, atol=1e-12)
    assert int((resc[:, T * n:] > 0).sum()) > 0 or B == 1            # some inequalities active


@pytest.mark.parametrize("B,T,k", [(5, 4, 1), (33, 20, 20), (2, 35, 3)])
def test_merit_kernel(B, T, k):
    """dqp_al_merit vs the torch restatement of al_utils.merit_function (al_utils.py:37-59), with
    and without a leading candidate axis (the 20-way line search)."""
    from diff_qp_mpc_amd import al_utils
    gen = torch.Generator().manual_seed(T + k)
    n, m = 2, 1
    shape = (k, B, T, n + m) if k > 1 else (B, T, n + m)
    xu = torch.randn(*shape, generator=gen, dtype=torch.float64).cuda()
    x0 = torch.randn(B, n, generator=gen, dtype=torch.float64).cuda()
    Q = (torch.rand(B, T, n + m, generator=gen, dtype=torch.float64) + 0.1).cuda()
    q = torch.randn(B, T, n + m, generator=gen, dtype=torch.float64).cuda()
    lam = torch.randn(B, T * n + 2 * T * m, generator=gen, dtype=torch.float64).cuda()
    rho = (10.0 ** torch.randint(0, 3, (B, 1), generator=gen).double()).cuda()
    lo, hi = torch.full((m,), -0.5, dtype=torch.float64).cuda(), torch.full((m,), 0.5, dtype=torch.float64).cuda()
    with torch.no_grad():
        got = al_utils.merit_function(xu, Q, q, Pendulum(), x0, lam, rho, None, None, lo, hi)
    with torch.enable_grad():          # grad mode on -> the torch path
        want = al_utils.merit_function(xu, Q, q, Pendulum(), x0, lam, rho, None, None, lo, hi)
    np.testing.assert_allclose(got.cpu().numpy(), want.detach().cpu().numpy(), rtol=1e-12, atol=1e-10)


@pytest.mark.parametrize("name", AL_CASES)
def test_al_mpc_two_calls_vs_reference(name):
    from diff_qp_mpc_amd import AL_mpc, al_utils
    g = load(name)
    B, T = g["in_Qd"].shape[:2]
    nx, nu = 2, 1
    x0 = dev(g["in_x0"])
    C = torch.diag_embed(dev(g["in_Qd"])).requires_grad_()
    c = dev(g["in_c"], grad=True)
    u_init = dev(g["in_u_init"])
    ctrl = AL_mpc.MPC(nx, nu, T, u_lower=dev(g["in_u_lower"]), u_upper=dev(g["in_u_upper"]), n_batch=B,
                      verbose=0, u_init=u_init, solver_type="dense", dtype=torch.float64, eps=1e-5,
                      exit_unconverged=False, backprop=False)
    ctrl.reinitialize(x0, torch.ones(B, T, 1, device="cuda"))
    ctrl.u_init = u_init
    dyn, dyn_jac = Pendulum(), PendulumJac()
    x, u = ctrl(x0, al_utils.QuadCost(C, c), dyn, dyn_jac)
    assert x.dtype == torch.float32 and u.dtype == torch.float32
    np.testing.assert_allclose(x.detach().cpu().numpy(), g["x1"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(u.detach().cpu().numpy(), g["u1"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ctrl.lamda_prev.cpu().numpy(), g["lam1"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ctrl.rho_prev.cpu().numpy(), g["rho1"], rtol=0, atol=0)
    (x.double().sum() + 2.0 * u.double().sum()).backward()
    np.testing.assert_allclose(C.grad.diagonal(dim1=-2, dim2=-1).cpu().numpy(), g["dC1"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(c.grad.cpu().numpy(), g["dc1"], rtol=1e-4, atol=1e-6)
    # second call: warm start from the stored history and the previous solution
    x2, u2 = ctrl(x0, al_utils.QuadCost(C.detach(), c.detach()), dyn, dyn_jac)
    np.testing.assert_allclose(x2.cpu().numpy(), g["x2"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(u2.cpu().numpy(), g["u2"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ctrl.lamda_prev.cpu().numpy(), g["lam2"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ctrl.rho_prev.cpu().numpy(), g["rho2"], rtol=0, atol=0)
