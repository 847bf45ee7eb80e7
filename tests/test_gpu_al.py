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


def _pin_lane_group(group):
    """one problem per 16-lane DPP row ("16") or per half row ("8", nt <= 8) in the block-tridiagonal kernels"""
    from diff_qp_mpc_amd import _lib
    assert _lib.load().dqp_al_lane_group(int(group)) == 0


@pytest.fixture(autouse=True)
def _auto_lane_group():
    yield
    from diff_qp_mpc_amd import _lib
    _lib.load().dqp_al_lane_group(0)


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


class LinearJac(torch.nn.Module):
    """x+ = A x + B u with m controls: exercises the [upper(m), lower(m)] row interleave for m > 1."""

    def __init__(self, n, m, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.A = (torch.eye(n, dtype=torch.float64) + 0.1 * torch.randn(n, n, generator=g, dtype=torch.float64)).cuda()
        self.B = torch.randn(n, m, generator=g, dtype=torch.float64).cuda()

    def step_np(self, x, u):
        A, Bm = self.A.cpu().numpy(), self.B.cpu().numpy()
        N = x.shape[0]
        return x @ A.T + u @ Bm.T, np.broadcast_to(A, (N,) + A.shape).copy(), np.broadcast_to(Bm, (N,) + Bm.shape).copy()

    def forward(self, x, u):
        xn = x @ self.A.T + u @ self.B.T
        N = x.shape[0]
        return xn, (self.A.expand(N, -1, -1), self.B.expand(N, -1, -1))


@pytest.mark.parametrize("B,T,n,m", [(5, 4, 2, 1), (33, 20, 2, 1), (1, 2, 2, 1), (7, 6, 3, 2), (4, 5, 4, 3)])
def test_assemble_jacobian_kernel(B, T, n, m):
    """dqp_al_assemble against the numpy oracle (oracle/al_oracle.py, pinned to the reference's
    ns_J / ns_Jc / ns_res by test_oracle_golden.py): exact for Jc (copies and +-1), 1e-12 for the
    products J^T lam + rho Jc^T res_c.  m > 1 exercises the [upper(m), lower(m)] row interleave."""
    from diff_qp_mpc_amd import al_utils
    gen = torch.Generator().manual_seed(T + 10 * m)
    xu = torch.randn(B, T, n + m, generator=gen, dtype=torch.float64).cuda()
    x0 = torch.randn(B, n, generator=gen, dtype=torch.float64).cuda()
    lam = torch.randn(B, T * n + 2 * T * m, generator=gen, dtype=torch.float64).cuda()
    rho = (10.0 ** torch.randint(0, 3, (B, 1), generator=gen).double()).cuda()
    lo, hi = torch.full((m,), -0.5, dtype=torch.float64).cuda(), torch.full((m,), 0.5, dtype=torch.float64).cuda()
    if m == 1:
        jac, step = PendulumJac(), al_oracle.pendulum_step
    else:
        jac = LinearJac(n, m, 3)
        step = jac.step_np
    res, resc, J, Jc_ref = al_oracle.constraint_jacobian(xu.cpu().numpy(), x0.cpu().numpy(), lo.cpu().numpy(),
                                                         hi.cpu().numpy(), step=step)
    g_ref = np.einsum("bc,bcn->bn", lam.cpu().numpy(), J) + rho.cpu().numpy() * np.einsum("bc,bcn->bn", resc, Jc_ref)
    Jc, gterm = al_utils.assemble_jacobian(xu, x0, jac, lam, rho, lo, hi)
    np.testing.assert_allclose(Jc.cpu().numpy(), Jc_ref, rtol=0, atol=1e-15)
    np.testing.assert_allclose(gterm.cpu().numpy(), g_ref, rtol=1e-12, atol=1e-12)
    assert int((resc[:, T * n:] > 0).sum()) > 0 or B == 1            # some inequalities active
    # and the package's own torch construction (the non-fused path) agrees with the oracle too
    _, _, J_t, Jc_t = al_utils.constraint_jacobian(xu, x0, jac, lo, hi)
    np.testing.assert_allclose(J_t.cpu().numpy(), J, rtol=0, atol=1e-15)
    np.testing.assert_allclose(Jc_t.cpu().numpy(), Jc_ref, rtol=0, atol=1e-15)


@pytest.mark.parametrize("name", AL_CASES)
def test_assemble_jacobian_kernel_vs_reference_golden(name):
    """The reference's own constraint Jacobian at the first iterate (ns_xu -> ns_Jc; al_utils.py:162-318)
    and its merit gradient (ns_grad, al_utils.py:62-102)."""
    from diff_qp_mpc_amd import al_utils
    g = load(name)
    xu, x0 = dev(g["ns_xu"]), dev(g["in_x0"])
    B = xu.shape[0]
    lam = torch.zeros(B, g["ns_Jc"].shape[1], dtype=torch.float64, device="cuda")
    rho = dev(g["ns_rho"])
    Jc, gterm = al_utils.assemble_jacobian(xu, x0, PendulumJac(), lam, rho, dev(g["in_u_lower"]), dev(g["in_u_upper"]))
    np.testing.assert_allclose(Jc.cpu().numpy(), g["ns_Jc"], rtol=0, atol=1e-13)
    grad = (dev(g["ns_Qd"]) * xu + dev(g["in_c"])).reshape(B, -1) + gterm
    np.testing.assert_allclose(grad.cpu().numpy(), g["ns_grad"], rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("B,T,k,n,m", [(5, 4, 1, 2, 1), (33, 20, 20, 2, 1), (2, 35, 3, 2, 1), (6, 5, 20, 3, 2)])
def test_merit_kernel(B, T, k, n, m):
    """dqp_al_merit against a numpy restatement of al_utils.merit_function (al_utils.py:37-59) built
    on the oracle's residuals, with and without a leading candidate axis (the 20-way line search)."""
    from diff_qp_mpc_amd import al_utils
    gen = torch.Generator().manual_seed(T + k)
    shape = (k, B, T, n + m) if k > 1 else (B, T, n + m)
    xu = torch.randn(*shape, generator=gen, dtype=torch.float64).cuda()
    x0 = torch.randn(B, n, generator=gen, dtype=torch.float64).cuda()
    Q = (torch.rand(B, T, n + m, generator=gen, dtype=torch.float64) + 0.1).cuda()
    q = torch.randn(B, T, n + m, generator=gen, dtype=torch.float64).cuda()
    lam = torch.randn(B, T * n + 2 * T * m, generator=gen, dtype=torch.float64).cuda()
    rho = (10.0 ** torch.randint(0, 3, (B, 1), generator=gen).double()).cuda()
    lo, hi = torch.full((m,), -0.5, dtype=torch.float64).cuda(), torch.full((m,), 0.5, dtype=torch.float64).cuda()
    if m == 1:
        dyn, step = Pendulum(), al_oracle.pendulum_step
    else:
        lj = LinearJac(n, m, 3)
        dyn, step = (lambda x, u: lj(x, u)[0]), lj.step_np
    with torch.no_grad():
        got = al_utils.merit_function(xu, Q, q, dyn, x0, lam, rho, None, None, lo, hi)
    xs = xu.cpu().numpy().reshape(-1, B, T, n + m)
    want = []
    for cand in xs:
        res, resc, _, _ = al_oracle.constraint_jacobian(cand, x0.cpu().numpy(), lo.cpu().numpy(), hi.cpu().numpy(), step=step)
        cost = (0.5 * cand * Q.cpu().numpy() * cand + q.cpu().numpy() * cand).sum((1, 2))
        want.append(cost + 0.5 * rho.cpu().numpy()[:, 0] * (resc * resc).sum(1) + (lam.cpu().numpy() * res).sum(1))
    np.testing.assert_allclose(got.cpu().numpy().reshape(-1, B), np.stack(want), rtol=1e-12, atol=1e-10)
    # a length-1 bound with m > 1 must not take the fused path (it would read past the bound array)
    if m > 1:
        with torch.no_grad():
            alt = al_utils.merit_function(xu, Q, q, dyn, x0, lam, rho, None, None, lo[:1], hi[:1])
        np.testing.assert_allclose(alt.cpu().numpy().reshape(-1, B), np.stack(want), rtol=1e-12, atol=1e-10)


@pytest.fixture(params=[0, 128], ids=["banded-jac", "dense-newton"])
def user_dynamics_path(request):
    """caller-supplied dynamics modules: the block-tridiagonal step on the module's Jacobians (default, every size) or the
    dense Newton step (dqp_al_assemble + dqp_al_newton_step, round 2's default below nz = 128)"""
    from diff_qp_mpc_amd import AL_mpc
    old = AL_mpc.BANDED_USER_DYNAMICS_FROM_NZ
    AL_mpc.BANDED_USER_DYNAMICS_FROM_NZ = request.param
    yield request.param
    AL_mpc.BANDED_USER_DYNAMICS_FROM_NZ = old


@pytest.mark.parametrize("name", AL_CASES)
def test_al_mpc_two_calls_vs_reference(name, user_dynamics_path):
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


@pytest.mark.parametrize("group", ["16", "8"])       # lanes per problem of the block-tridiagonal kernels (8: nt <= 8 models)
@pytest.mark.parametrize("name,robot", [("CFG3_cartpole1l_T20_b4", "cartpole1l"), ("CFG5_cartpole2l_T5_b4", "cartpole2l"),
                                        ("CFG4_rexquadrotor_T30_b4", "rexquadrotor"),
                                        ("CFG4_rexquadrotor_T6_b4", "rexquadrotor")])
def test_al_mpc_cartpole_vs_reference(name, robot, group, monkeypatch):
    """BASELINE config 3 (cartpole-1, n 4, m 1, T 20), the config-5 robot (cartpole-2, n 6, T 5) and
    config 4 (quadrotor, n 12, m 4, T 30: nz = 480, and a T = 6 case)
    through AL_mpc.MPC with the DEVICE dynamics registry, against the reference's AL_mpc.MPC run on
    its own dynamics (CasADi-generated C: make_golden_cfg3.py; rex_quadrotor.py: make_golden_cfg4.py):
    cold call with gradients, then the warm-started call.  x, u are float32 in the reference (AL_mpc.py:319-320): rtol 1e-4 / atol 1e-4
    (states reach +-pi, controls +-100); multipliers rtol 1e-5 / atol 1e-5; rho exact."""
    from diff_qp_mpc_amd import AL_mpc, al_utils
    from diff_qp_mpc_amd.dynamics import DeviceDynamics
    _pin_lane_group(group)
    g = load(name)
    B, T = g["in_Qd"].shape[:2]
    dyn = DeviceDynamics(robot, dt=float(g["dt"]))
    nx, nu = dyn.n_state, dyn.n_ctrl
    x0 = dev(g["in_x0"])
    C = torch.diag_embed(dev(g["in_Qd"])).requires_grad_()
    c = dev(g["in_c"], grad=True)
    ctrl = AL_mpc.MPC(nx, nu, T, u_lower=dev(g["in_u_lower"]), u_upper=dev(g["in_u_upper"]), n_batch=B,
                      verbose=0, solver_type="dense", dtype=torch.float64, eps=1e-5,
                      exit_unconverged=False, backprop=False)
    ctrl.reinitialize(x0, torch.ones(B, T, 1, device="cuda"))
    ctrl.x_init, ctrl.u_init = dev(g["in_x_init"]), dev(g["in_u_init"])
    x, u = ctrl(x0, al_utils.QuadCost(C, c), dyn, dyn.jac)
    np.testing.assert_allclose(x.detach().cpu().numpy(), g["x1"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(u.detach().cpu().numpy(), g["u1"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(ctrl.lamda_prev.cpu().numpy(), g["lam1"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ctrl.rho_prev.cpu().numpy(), g["rho1"], rtol=0, atol=0)
    (x.double().sum() + 2.0 * u.double().sum()).backward()
    np.testing.assert_allclose(C.grad.diagonal(dim1=-2, dim2=-1).cpu().numpy(), g["dC1"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(c.grad.cpu().numpy(), g["dc1"], rtol=1e-4, atol=1e-5)
    x2, u2 = ctrl(x0, al_utils.QuadCost(C.detach(), c.detach()), dyn, dyn.jac)
    np.testing.assert_allclose(x2.cpu().numpy(), g["x2"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(u2.cpu().numpy(), g["u2"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(ctrl.lamda_prev.cpu().numpy(), g["lam2"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ctrl.rho_prev.cpu().numpy(), g["rho2"], rtol=0, atol=0)


def test_config3_full_size_properties():
    """B = 4096, n 4, m 1, T = 20 (BASELINE config 3 size): one AL_mpc.MPC call on the device
    cartpole.  Size-independent properties: x_0 pinned to x0 (al_utils.py:515), finite outputs,
    multipliers of the inequality block non-negative (AL_mpc.py:300-301), rho = 100 after two outer
    iterations of a cold start (AL_mpc.py:307), and batch-size independence against the reference:
    the first four problems are the golden fixture's."""
    from diff_qp_mpc_amd import AL_mpc, al_utils
    from diff_qp_mpc_amd.dynamics import DeviceDynamics
    B, T = 4096, 20
    dyn = DeviceDynamics("cartpole1l", dt=0.05)
    nx, nu = 4, 1
    rng = np.random.default_rng(0)
    x0 = dev(rng.uniform(-np.pi, np.pi, (B, nx)))
    Qd = torch.cat([torch.ones(nx), 1e-8 * torch.ones(nu)]).double().cuda().repeat(B, T, 1)
    ramp = torch.linspace(1.0, 0.0, T, dtype=torch.float64, device="cuda")[None, :, None]
    x_ref = x0[:, None, :] * ramp
    xu_ref = torch.cat([x_ref, torch.zeros(B, T, nu, dtype=torch.float64, device="cuda")], -1)
    C = torch.diag_embed(Qd)
    c = -(Qd * xu_ref)
    lo, hi = dev([-100.0]), dev([100.0])
    ctrl = AL_mpc.MPC(nx, nu, T, u_lower=lo, u_upper=hi, n_batch=B, verbose=0, solver_type="dense",
                      dtype=torch.float64, eps=1e-5, exit_unconverged=False, backprop=False)
    ctrl.reinitialize(x0, torch.ones(B, T, 1, device="cuda"))
    ctrl.x_init, ctrl.u_init = x_ref.clone(), torch.zeros(B, T, nu, dtype=torch.float64, device="cuda")
    x, u = ctrl(x0, al_utils.QuadCost(C, c), dyn, dyn.jac)
    assert x.shape == (B, T, nx) and u.shape == (B, T, nu)
    assert bool(torch.isfinite(x).all()) and bool(torch.isfinite(u).all())
    assert float((x[:, 0].double() - x0).abs().max()) < 1e-6           # float32 output of an exact pin
    assert float(ctrl.lamda_prev[:, T * nx:].min()) >= 0.0
    assert set(ctrl.rho_prev.unique().tolist()) <= {100.0}
    # samples are independent: the first four problems are the golden's (same generator stream), so
    # inside the 4096-batch they must come out as the reference computed them in a batch of four
    g = load("CFG3_cartpole1l_T20_b4")
    np.testing.assert_allclose(x0[:4].cpu().numpy(), g["in_x0"], rtol=0, atol=0)
    np.testing.assert_allclose(x[:4].cpu().numpy(), g["x1"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(u[:4].cpu().numpy(), g["u1"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(ctrl.lamda_prev[:4].cpu().numpy(), g["lam1"], rtol=1e-5, atol=1e-5)


def test_config4_full_size_properties():
    """BASELINE config 4 at its size: quadrotor n 12, m 4, T = 30 (nz = 480), B = 8192 -- one
    AL_mpc.MPC call through the block-tridiagonal device path.  Size-independent properties as for
    config 3, and batch-size independence against the reference: the first four problems are the
    golden fixture's (make_golden_cfg4.py), solved here inside the 8192-batch."""
    from diff_qp_mpc_amd import AL_mpc, al_utils
    from diff_qp_mpc_amd.dynamics import DeviceDynamics
    g = load("CFG4_rexquadrotor_T30_b4")
    B, T, nx, nu = 8192, 30, 12, 4
    dyn = DeviceDynamics("rexquadrotor", dt=float(g["dt"]))
    rng = np.random.default_rng(0)
    win = np.array([1.0] * 3 + [0.15] * 3 + [0.5] * 3 + [0.25] * 3)
    x0 = dev(rng.uniform(-1, 1, (B, nx)) * win)
    np.testing.assert_allclose(x0[:4].cpu().numpy(), g["in_x0"], rtol=0, atol=0)      # same generator stream
    Qd = dev(g["in_Qd"][0, 0]).repeat(B, T, 1)
    ramp = torch.linspace(1.0, 0.0, T, dtype=torch.float64, device="cuda")[None, :, None]
    x_ref = x0[:, None, :] * ramp
    u_ref = dev(g["in_u_init"][0, 0]).repeat(B, T, 1)
    C = torch.diag_embed(Qd)
    c = -(Qd * torch.cat([x_ref, u_ref], -1))
    np.testing.assert_allclose(c[:4].cpu().numpy(), g["in_c"], rtol=0, atol=1e-15)
    ctrl = AL_mpc.MPC(nx, nu, T, u_lower=dev(g["in_u_lower"]), u_upper=dev(g["in_u_upper"]), n_batch=B, verbose=0,
                      solver_type="dense", dtype=torch.float64, eps=1e-5, exit_unconverged=False, backprop=False)
    ctrl.reinitialize(x0, torch.ones(B, T, 1, device="cuda"))
    ctrl.x_init, ctrl.u_init = x_ref.clone(), u_ref.clone()
    x, u = ctrl(x0, al_utils.QuadCost(C, c), dyn, dyn.jac)
    assert x.shape == (B, T, nx) and u.shape == (B, T, nu)
    assert bool(torch.isfinite(x).all()) and bool(torch.isfinite(u).all())
    assert float((x[:, 0].double() - x0).abs().max()) < 1e-6
    assert float(ctrl.lamda_prev[:, T * nx:].min()) >= 0.0
    assert set(ctrl.rho_prev.unique().tolist()) <= {100.0}
    np.testing.assert_allclose(x[:4].cpu().numpy(), g["x1"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(u[:4].cpu().numpy(), g["u1"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(ctrl.lamda_prev[:4].cpu().numpy(), g["lam1"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("group", ["16", "8"])
@pytest.mark.parametrize("robot,T", [("pendulum_euler", 20), ("cartpole1l", 20), ("cartpole2l", 5), ("pendulum_dx", 10),
                                     ("pendulum1l", 3), ("rexquadrotor", 30), ("rexquadrotor", 4)])
def test_banded_newton_step_vs_dense_oracle(robot, T, group, monkeypatch):
    """dqp_al_banded_newton_step (block-tridiagonal Cholesky, every knot in registers) against the
    reference's dense formulation restated in numpy (oracle/al_oracle.py: dense constraint Jacobian,
    H = diag(Q) + rho Jc^T Jc, dense Cholesky solve) with the device model's own Jacobians: the same
    matrix, factored in block form -> update rtol 1e-8 / atol 1e-10; and dqp_al_banded_solve against
    the dense chol solve."""
    import ctypes
    from diff_qp_mpc_amd import _lib, al_utils
    from diff_qp_mpc_amd.dynamics import DeviceDynamics
    _pin_lane_group(group)      # one problem per 16-lane row, or per half row (nt <= 8)
    lib = _lib.load()
    dyn = DeviceDynamics(robot)
    n, m, nt = dyn.n_state, dyn.n_ctrl, dyn.n_state + dyn.n_ctrl
    B = 37
    gen = torch.Generator().manual_seed(T)
    rnd = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float64).cuda()
    xu = 0.5 * rnd(B, T, nt)
    if robot == "rexquadrotor":
        xu[..., n:] = 14.5 + 0.3 * rnd(B, T, m)          # around hover; bounds below are +-0.3 about it
    if robot == "pendulum_dx":
        xu[..., :2] = torch.nn.functional.normalize(xu[..., :2], dim=-1)
    x0 = xu[:, 0, :n] + 0.1 * rnd(B, n)
    Qd = torch.rand(B, T, nt, generator=gen, dtype=torch.float64).cuda() + 0.1
    q = rnd(B, T, nt)
    ncon = T * n + 2 * T * m
    lam = rnd(B, ncon)
    rho = (10.0 ** torch.randint(0, 3, (B, 1), generator=gen).double()).cuda()
    mid = 14.5 if robot == "rexquadrotor" else 0.0
    lo, hi = torch.full((m,), mid - 0.3, dtype=torch.float64).cuda(), torch.full((m,), mid + 0.3, dtype=torch.float64).cuda()

    def step_np(x, u):
        xn, (Jx, Ju) = dyn.jac(dev(x), dev(u))
        return xn.cpu().numpy(), Jx.cpu().numpy(), Ju.cpu().numpy()

    res, resc, J, Jc = al_oracle.constraint_jacobian(xu.cpu().numpy(), x0.cpu().numpy(), lo.cpu().numpy(),
                                                     hi.cpu().numpy(), step=step_np)
    assert (resc[:, T * n:] > 0).any()
    grad = al_oracle.merit_grad(xu.cpu().numpy(), Qd.cpu().numpy(), q.cpu().numpy(), lam.cpu().numpy(),
                                rho.cpu().numpy(), resc, J, Jc)
    upd_ref, L_ref, info_ref = al_oracle.newton_update(Jc, Qd.cpu().numpy().reshape(B, -1), rho.cpu().numpy(), grad)
    assert not info_ref.any()
    dims = _lib.dqp_al_mpc_dims(B, n, m, T)
    fac = torch.empty(int(lib.dqp_al_banded_factor_bytes(ctypes.byref(dims), dyn.id)) // 8, dtype=torch.float64, device="cuda")
    upd = torch.empty(B, T, nt, dtype=torch.float64, device="cuda")
    info = torch.empty(B, dtype=torch.int32, device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = lib.dqp_al_banded_newton_step(ctypes.byref(dims), dyn.id, dyn.dt, P(xu), P(x0), P(Qd), P(q), P(lam),
                                       P(rho.reshape(B).contiguous()), P(lo), P(hi), P(upd), P(fac), P(info), None)
    assert rc == 0
    torch.cuda.synchronize()
    assert int(info.abs().max()) == 0
    np.testing.assert_allclose(upd.cpu().numpy().reshape(B, -1), upd_ref, rtol=1e-8, atol=1e-10)
    rhs = rnd(B, T, nt)
    out = torch.empty_like(rhs)
    assert lib.dqp_al_banded_solve(ctypes.byref(dims), dyn.id, P(fac), P(rhs), P(out), None) == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy().reshape(B, -1),
                               al_oracle.chol_solve_neg(L_ref, rhs.cpu().numpy().reshape(B, -1)), rtol=1e-8, atol=1e-10)


def test_reference_entry_points_of_the_jacobian_fill():
    """al_utils.constraint_res_jac2 / dyn_res_eq_jac / dyn_res_ineq_jac / merit_hessian under the
    reference's names and return conventions (al_utils.py:105-186,212-318), against the numpy oracle."""
    from diff_qp_mpc_amd import al_utils
    from diff_qp_mpc_amd.dynamics import DeviceDynamics
    dyn = DeviceDynamics("cartpole1l")
    B, T, n, m = 5, 6, 4, 1
    gen = torch.Generator().manual_seed(1)
    xu = 0.5 * torch.randn(B, T, n + m, generator=gen, dtype=torch.float64).cuda()
    x0 = xu[:, 0, :n] + 0.1
    lo, hi = dev([-0.3]), dev([0.3])
    res, resc, J, Jc, H = al_utils.constraint_res_jac2(xu, x0, dyn.jac, None, None, lo, hi)

    def step_np(x, u):
        xn, (Jx, Ju) = dyn.jac(dev(x), dev(u))
        return xn.cpu().numpy(), Jx.cpu().numpy(), Ju.cpu().numpy()
    r0, rc0, J0, Jc0 = al_oracle.constraint_jacobian(xu.cpu().numpy(), x0.cpu().numpy(), lo.cpu().numpy(), hi.cpu().numpy(), step=step_np)
    np.testing.assert_allclose(res.cpu().numpy(), r0, atol=1e-13)
    np.testing.assert_allclose(resc.cpu().numpy(), rc0, atol=1e-13)
    np.testing.assert_allclose(J.cpu().numpy(), J0, atol=1e-12)
    np.testing.assert_allclose(Jc.cpu().numpy(), Jc0, atol=1e-12)
    np.testing.assert_allclose(H.cpu().numpy(), Jc0.transpose(0, 2, 1) @ Jc0, atol=1e-11)
    x, u = xu[..., :n], xu[..., n:]
    re, Je = al_utils.dyn_res_eq_jac(x, u, dyn.jac, x0)
    np.testing.assert_allclose(re.cpu().numpy(), r0[:, :T * n], atol=1e-13)
    np.testing.assert_allclose(Je.cpu().numpy(), J0[:, :T * n], atol=1e-12)
    ri, ric, Ji, Jic = al_utils.dyn_res_ineq_jac(x, u, x0, None, None, lo, hi)
    np.testing.assert_allclose(Ji.cpu().numpy(), J0[:, T * n:], atol=0)
    np.testing.assert_allclose(Jic.cpu().numpy(), Jc0[:, T * n:], atol=0)
    Qd = torch.rand(B, T, n + m, generator=gen, dtype=torch.float64).cuda() + 0.1
    rho = dev([[1.0], [10.0], [100.0], [1.0], [10.0]])
    Hm = al_utils.merit_hessian(xu, Qd, None, dyn.jac, x0, None, rho, None, None, lo, hi)
    want = np.stack([np.diag(Qd[b].reshape(-1).cpu().numpy()) + float(rho[b]) * Jc0[b].T @ Jc0[b] for b in range(B)])
    np.testing.assert_allclose(Hm.cpu().numpy(), want, atol=1e-10)


class _UserModule(torch.nn.Module):
    """A dynamics the solver knows nothing about (a plain callable with its own Jacobian function), as a
    caller's torch module would be; the arithmetic is borrowed from the registry so that the reference
    fixtures apply."""

    def __init__(self, name, dt):
        super().__init__()
        from diff_qp_mpc_amd.dynamics import DeviceDynamics
        self._d = DeviceDynamics(name, dt=dt)

    def forward(self, x, u):
        return self._d(x, u)

    def jac(self, x, u):
        return self._d.jac(x, u)


@pytest.mark.parametrize("group", ["16", "8"])
@pytest.mark.parametrize("name,robot", [("CFG4_rexquadrotor_T30_b4", "rexquadrotor"), ("CFG4_rexquadrotor_T6_b4", "rexquadrotor"),
                                        ("CFG3_cartpole1l_T20_b4", "cartpole1l")])
def test_al_mpc_user_dynamics_module_banded(name, robot, group, monkeypatch):
    """AL_mpc.MPC with a caller-supplied dynamics module (not a DeviceDynamics): from nz > 128 on its own
    Jacobians go to dqp_al_banded_newton_step_jac (config 4: nz = 480, beyond the dense Newton step).  Against
    the reference's AL_mpc.MPC fixtures (cold call, gradients, warm-started call); the cartpole case (nz = 100)
    forces the banded path below the threshold and must agree with the same fixtures too."""
    _pin_lane_group(group)
    from diff_qp_mpc_amd import AL_mpc, al_utils
    g = load(name)
    B, T = g["in_Qd"].shape[:2]
    dyn = _UserModule(robot, float(g["dt"])).cuda()
    nx, nu = dyn._d.n_state, dyn._d.n_ctrl
    old = AL_mpc.BANDED_USER_DYNAMICS_FROM_NZ
    AL_mpc.BANDED_USER_DYNAMICS_FROM_NZ = 0
    try:
        x0 = dev(g["in_x0"])
        C = torch.diag_embed(dev(g["in_Qd"])).requires_grad_()
        c = dev(g["in_c"], grad=True)
        ctrl = AL_mpc.MPC(nx, nu, T, u_lower=dev(g["in_u_lower"]), u_upper=dev(g["in_u_upper"]), n_batch=B,
                          verbose=0, solver_type="dense", dtype=torch.float64, eps=1e-5,
                          exit_unconverged=False, backprop=False)
        ctrl.reinitialize(x0, torch.ones(B, T, 1, device="cuda"))
        ctrl.x_init, ctrl.u_init = dev(g["in_x_init"]), dev(g["in_u_init"])
        x, u = ctrl(x0, al_utils.QuadCost(C, c), dyn, dyn.jac)
        np.testing.assert_allclose(x.detach().cpu().numpy(), g["x1"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(u.detach().cpu().numpy(), g["u1"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(ctrl.lamda_prev.cpu().numpy(), g["lam1"], rtol=1e-5, atol=1e-5)
        (x.double().sum() + 2.0 * u.double().sum()).backward()
        np.testing.assert_allclose(C.grad.diagonal(dim1=-2, dim2=-1).cpu().numpy(), g["dC1"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(c.grad.cpu().numpy(), g["dc1"], rtol=1e-4, atol=1e-5)
        x2, u2 = ctrl(x0, al_utils.QuadCost(C.detach(), c.detach()), dyn, dyn.jac)
        np.testing.assert_allclose(x2.cpu().numpy(), g["x2"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(u2.cpu().numpy(), g["u2"], rtol=1e-4, atol=1e-4)
    finally:
        AL_mpc.BANDED_USER_DYNAMICS_FROM_NZ = old


@pytest.mark.parametrize("name,robot", [("CFG5_cartpole2l_T5_b4", "cartpole2l"), ("CFG4_rexquadrotor_T30_b4", "rexquadrotor")])
def test_al_mpc_cholesky_failure_takes_the_lu_path(name, robot):
    """A merit Hessian that is not positive definite (one sample's control cost made strongly negative): the
    reference's NewtonAL sees a NaN Cholesky update and switches the batch to an LU solve, this step and after
    (al_utils.py:419-427), backward included (:468-472).  AL_mpc.MPC on the device path notices the failure flag,
    redoes the solve on the general path -- dense Newton kernel below nz = 128 (cartpole-2: nz 35), straight LU above
    it (quadrotor T = 30: nz 480, beyond dqp_al_newton_step) -- and must agree with the numpy oracle of the whole
    call (oracle/al_solve_oracle.py, pinned by the same fixtures): x, u rtol 1e-4 / atol 1e-4, gradients of the
    healthy samples rtol 1e-3 / atol 1e-5 (an indefinite Newton system is badly conditioned)."""
    from diff_qp_mpc_amd import AL_mpc, al_utils
    from diff_qp_mpc_amd.dynamics import DeviceDynamics
    from oracle import al_solve_oracle as aso, dyn_host
    g = load(name)
    B, T = g["in_Qd"].shape[:2]
    dyn = DeviceDynamics(robot, dt=float(g["dt"]))
    nx, nu = dyn.n_state, dyn.n_ctrl
    Qd = g["in_Qd"].copy()
    Qd[1, :, nx:] = -5.0e3                                  # sample 1: negative control cost -> indefinite Hessian
    step = dyn_host.stepper(robot, float(g["dt"]))
    if step is None:
        pytest.skip("hipcc not available for the host build of the dynamics")
    lam0, rho0 = np.zeros((B, T * nx + 2 * T * nu)), np.ones((B, 1))
    o = aso.al_solve(g["in_x_init"], g["in_u_init"], g["in_x0"], Qd, g["in_c"], g["in_u_lower"], g["in_u_upper"], step,
                     lam0, rho0)
    assert o["chol_fail"]
    x0 = dev(g["in_x0"])
    C = torch.diag_embed(dev(Qd)).requires_grad_()
    c = dev(g["in_c"], grad=True)
    ctrl = AL_mpc.MPC(nx, nu, T, u_lower=dev(g["in_u_lower"]), u_upper=dev(g["in_u_upper"]), n_batch=B, verbose=0,
                      solver_type="dense", dtype=torch.float64, eps=1e-5, exit_unconverged=False, backprop=False)
    ctrl.reinitialize(x0, torch.ones(B, T, 1, device="cuda"))
    ctrl.x_init, ctrl.u_init = dev(g["in_x_init"]), dev(g["in_u_init"])
    x, u = ctrl(x0, al_utils.QuadCost(C, c), dyn, dyn.jac)
    ok = np.array([0, 2, 3])
    np.testing.assert_allclose(x.detach().cpu().numpy()[ok], o["x"][ok], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(u.detach().cpu().numpy()[ok], o["u"][ok], rtol=1e-4, atol=1e-4)
    assert bool(torch.isfinite(x).all()) and bool(torch.isfinite(u).all())
    (x.double().sum() + 2.0 * u.double().sum()).backward()
    gxu = np.concatenate((np.ones((B, T, nx)), 2.0 * np.ones((B, T, nu))), 2)
    dQ, dq = aso.backward(o["L"], o["xu"], gxu, chol_fail=True)
    np.testing.assert_allclose(C.grad.diagonal(dim1=-2, dim2=-1).cpu().numpy()[ok], dQ[ok], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(c.grad.cpu().numpy()[ok], dq[ok], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("robot,T,B", [("cartpole2l", 5, 128), ("rexquadrotor", 6, 37)])
def test_al_graphed_mpc_bitwise_equal_to_eager(robot, T, B):
    """AL_mpc.GraphedMPC (a cold AL_mpc.MPC call -- reinitialize + forward -- and its backward replayed as hipGraphs)
    against the eager call on the same inputs: x, u and the gradients wrt C and c bit for bit, on the captured batch and
    on a second batch copied into the static inputs; and the one-call solve (dqp_al_mpc_solve) against round 2's host loop
    around the Newton solves (ONE_CALL_SOLVE = False): identical kernels in the same order, bit for bit as well."""
    from diff_qp_mpc_amd import AL_mpc, al_utils
    from diff_qp_mpc_amd.dynamics import DeviceDynamics
    dyn = DeviceDynamics(robot)
    nx, nu = dyn.n_state, dyn.n_ctrl
    rng = np.random.default_rng(3)
    lo, hi = (dev(np.full(nu, 11.5)), dev(np.full(nu, 18.3))) if robot == "rexquadrotor" else (dev(np.full(nu, -250.0)), dev(np.full(nu, 250.0)))
    Qd = dev(np.concatenate([np.ones(nx), 1e-3 * np.ones(nu)])).repeat(B, T, 1)

    def batch(seed):
        r = np.random.default_rng(seed)
        x0 = dev(r.uniform(-0.5, 0.5, (B, nx)))
        x_ref = x0[:, None, :] * torch.linspace(1.0, 0.0, T, dtype=torch.float64, device="cuda")[None, :, None]
        u_ref = ((lo + hi) / 2).repeat(B, T, 1)
        C = torch.diag_embed(Qd).requires_grad_()
        c = (-(Qd * torch.cat([x_ref, u_ref], -1))).clone().requires_grad_()
        return x0, x_ref, u_ref, C, c

    def make():
        return AL_mpc.MPC(nx, nu, T, u_lower=lo, u_upper=hi, n_batch=B, verbose=0, solver_type="dense", dtype=torch.float64,
                          eps=1e-5, exit_unconverged=False, backprop=False)

    def eager(x0, x_ref, u_ref, C, c):
        ctrl = make()
        ctrl.reinitialize(x0, torch.ones(B, T, 1, device="cuda"))
        ctrl.x_init, ctrl.u_init = x_ref, u_ref
        x, u = ctrl(x0, al_utils.QuadCost(C, c), dyn, dyn.jac)
        gC, gc = torch.autograd.grad(x.double().sum() + 2.0 * u.double().sum(), (C, c))
        return x.detach(), u.detach(), gC, gc

    x0, x_ref, u_ref, C, c = batch(0)
    want = eager(x0, x_ref, u_ref, C, c)
    AL_mpc.ONE_CALL_SOLVE = False
    try:
        old = eager(x0, x_ref, u_ref, C, c)
    finally:
        AL_mpc.ONE_CALL_SOLVE = True
    for a, b in zip(old, want):
        assert torch.equal(a, b)
    ctrl = make()
    ctrl.mask = torch.ones(B, T, 1, device="cuda")
    g = AL_mpc.GraphedMPC(ctrl, (x0, C, c), dyn, x_init=x_ref, u_init=u_ref)
    for seed in (0, 1):
        x0b, x_refb, u_refb, Cb, cb = batch(seed)
        g.x_init.copy_(x_refb); g.u_init.copy_(u_refb)         # the static initial guess of the captured call
        x, u = g(x0b, Cb, cb)
        gC, gc = torch.autograd.grad(x.double().sum() + 2.0 * u.double().sum(), (Cb, cb))
        wb = eager(x0b, x_refb, u_refb, Cb, cb)
        for a, b in zip((x, u, gC, gc), wb):
            assert torch.equal(a, b)
    assert not g.failed()
