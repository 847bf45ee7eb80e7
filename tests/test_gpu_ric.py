"""Stage-wise (Riccati) PDIPM for MPC-structured QPs (csrc/dqp_ric.hip; SURVEY.md §8 f1 / row g,
BASELINE config 4 shape n 12, m 4, T 30) through the C ABI (dqp_mpc_qp_forward / _backward).

Checkers: (1) the assemble -> DenseQPFunction pipeline of this package on the dense GPU kernels
(themselves pinned to the reference goldens) where the dense QP still fits them (nz <= 64);
(2) the CPU oracle's restatement of the reference's DenseQPFunction (batch_LU.py, what
qp_wrapper.MPC calls) on the QP assembled in numpy, at sizes no dense GPU kernel covers;
(3) the reference's own qp_wrapper.MPC outputs at n 12, m 4, T 6 (tests/golden/make_golden_ric.py);
(4) size-independent KKT properties at the full config-4 size.
Tolerances as in test_gpu_parity.py: zhat rtol 1e-6 / atol 1e-8, duals rtol 1e-5 / atol 1e-7,
gradients rtol 1e-4 / atol 1e-6.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import oracle

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
ZT = dict(rtol=1e-6, atol=1e-8)
DT = dict(rtol=1e-5, atol=1e-7)
GT = dict(rtol=1e-4, atol=1e-6)


def dev(a, grad=False):
    t = torch.tensor(np.asarray(a), dtype=torch.float64, device="cuda")
    return t.requires_grad_() if grad else t


def problem(n, m, T, B, seed, active=0.4, spread=0.15):
    """MPC data in the reference's time-major layout (qp_wrapper.py:124-160): SPD stage costs, mildly
    unstable linear dynamics, bounds tight enough that a good share of the controls sit on them."""
    rng = np.random.default_rng(seed)
    nt = n + m
    L = rng.standard_normal((T, B, nt, nt)) * 0.3
    C = L @ L.transpose(0, 1, 3, 2) + np.eye(nt)
    c = rng.standard_normal((T, B, nt))
    F = np.concatenate([np.eye(n) + spread * rng.standard_normal((T - 1, B, n, n)),
                        0.5 * rng.standard_normal((T - 1, B, n, m))], axis=-1)
    f = 0.1 * rng.standard_normal((T - 1, B, n))
    x0 = rng.standard_normal((B, n))
    return C, c, F, f, x0, -active * np.ones(m), active * np.ones(m)


def assemble(C, c, F, f, x0, lo, hi):
    """The dense QP of qp_wrapper.py:638-679 in numpy (reference orderings)."""
    T, B, nt, _ = C.shape
    n = x0.shape[1]
    m = nt - n
    nz, neq, nineq = T * nt, T * n, 2 * T * m
    Q = np.zeros((B, nz, nz)); p = np.zeros((B, nz)); A = np.zeros((B, neq, nz)); b = np.zeros((B, neq))
    G = np.zeros((B, nineq, nz)); h = np.zeros((B, nineq))
    for t in range(T):
        Q[:, t * nt:(t + 1) * nt, t * nt:(t + 1) * nt] = C[t]
        p[:, t * nt:(t + 1) * nt] = c[t]
        for a in range(m):
            G[:, t * m + a, t * nt + n + a] = 1.0; h[:, t * m + a] = hi[a]
            G[:, T * m + t * m + a, t * nt + n + a] = -1.0; h[:, T * m + t * m + a] = -lo[a]
    for t in range(T - 1):
        A[:, t * n:(t + 1) * n, t * nt:(t + 1) * nt] = F[t]
        A[:, t * n:(t + 1) * n, (t + 1) * nt:(t + 1) * nt + n] = -np.eye(n)
        b[:, t * n:(t + 1) * n] = -f[t]
    A[:, (T - 1) * n:, :n] = np.eye(n); b[:, (T - 1) * n:] = x0
    return Q, p, G, h, A, b


def run_fused(n, m, T, data, w=None):
    from diff_qp_mpc_amd import qp_wrapper
    C, c, F, f, x0, lo, hi = data
    B = x0.shape[0]
    assert qp_wrapper._MPCQP.supported(B, n, m, T)
    ins = [dev(a, grad=True) for a in (C, c, F, f, x0)]
    tau = qp_wrapper._MPCQP.apply(*ins, dev(lo), dev(hi), n, m, T)
    if w is None:
        w = torch.linspace(0.5, 1.5, tau.numel(), dtype=torch.float64, device="cuda").reshape(tau.shape)
    (tau * w).sum().backward()
    return tau.detach().cpu().numpy(), [t.grad.cpu().numpy() for t in ins], w.cpu().numpy()


@pytest.mark.parametrize("n,m,T,B", [(3, 3, 4, 37), (4, 2, 5, 9), (12, 4, 4, 6), (2, 1, 7, 5)])
def test_stagewise_equals_assemble_plus_dense_gpu(n, m, T, B):
    """Same problems through the stage-wise kernels and through dqp_mpc_assemble + DenseQPFunction on
    the dense GPU kernels (nz <= 64), per-problem termination on both sides so that every QP is
    compared at its own converged iterate."""
    from diff_qp_mpc_amd import qp as qpmod, qp_wrapper
    data = problem(n, m, T, B, seed=n * 100 + T)
    old = qpmod.TERMINATION
    qpmod.TERMINATION = "per_problem"
    try:
        tau, grads, w = run_fused(n, m, T, data)
        C, c, F, f, x0, lo, hi = data
        ins = [dev(a, grad=True) for a in (C, c, F, f, x0)]
        Q, p, G, h, A, b = qp_wrapper._AssembleDenseQP.apply(*ins, dev(lo), dev(hi), n, m, T)
        z = qpmod.DenseQPFunction(verbose=-1)(Q, p, G, h, A, b)
        (z.reshape(B, T, n + m) * dev(w)).sum().backward()
    finally:
        qpmod.TERMINATION = old
    np.testing.assert_allclose(tau.reshape(B, -1), z.detach().cpu().numpy(), **ZT)
    for a, t, k in zip(grads, ins, ("dC", "dc", "dF", "df", "dx0")):
        np.testing.assert_allclose(a, t.grad.cpu().numpy(), err_msg=k, **GT)


@pytest.mark.parametrize("n,m,T,B", [(12, 4, 6, 5), (3, 1, 14, 8), (12, 4, 12, 3)])
def test_stagewise_vs_cpu_oracle(n, m, T, B):
    """Against the CPU oracle's DenseQPFunction restatement (batch_LU.py:29-244, qp.py:239-270) on the
    numpy-assembled QP: batch-coupled termination on both sides (the default)."""
    data = problem(n, m, T, B, seed=7 * n + T)
    tau, grads, w = run_fused(n, m, T, data)
    Q, p, G, h, A, b = assemble(*data)
    o = oracle.dense_forward(Q, p, G, h, A, b)
    np.testing.assert_allclose(tau.reshape(B, -1), o["zhat"], **ZT)
    og = oracle.dense_backward(o["K"], o["zhat"], o["lam"], o["nu"], w.reshape(B, -1))
    nt = n + m
    dC = np.stack([og["dQ"][:, t * nt:(t + 1) * nt, t * nt:(t + 1) * nt] for t in range(T)])
    dc = np.stack([og["dp"][:, t * nt:(t + 1) * nt] for t in range(T)])
    dF = np.stack([og["dA"][:, t * n:(t + 1) * n, t * nt:(t + 1) * nt] for t in range(T - 1)])
    df = np.stack([-og["db"][:, t * n:(t + 1) * n] for t in range(T - 1)])
    dx0 = og["db"][:, (T - 1) * n:]
    for a, want, k in zip(grads, (dC, dc, dF, df, dx0), ("dC", "dc", "dF", "df", "dx0")):
        np.testing.assert_allclose(a, want, err_msg=k, **GT)


@pytest.mark.parametrize("B,batch_rule", [(256, False), (8192, True)])
def test_config4_size_kkt_properties(B, batch_rule):
    """n 12, m 4, T 30 (nz 480, nineq 240, neq 360) at B = 256 and at BASELINE config 4's own batch, B = 8192 (771 MB of
    workspace, the kernels' 32-bit offset guards; there under the batch rule, with its 1.4 GB termination buffer):
    stationarity, primal feasibility, complementarity and sign conditions of the returned (tau, lam, nu, slack) on the
    original data."""
    from diff_qp_mpc_amd import _lib
    n, m, T = 12, 4, 30
    nt = n + m
    C, c, F, f, x0, lo, hi = problem(n, m, T, B, seed=3, spread=0.05)      # 30 steps: keep the rollout bounded
    lib = _lib.load()
    dims = _lib.dqp_mpc_dims(B, n, m, T, 1, 0)
    assert lib.dqp_mpc_qp_supported(ctypes.byref(dims)) == 1
    opts = _lib.dqp_opts(1e-12, 1e-10, 20, 3, _lib.DQP_FLAG_BATCH_TERMINATION if batch_rule else 0, 0)
    t = [dev(a) for a in (C, c, F, f, x0, lo, hi)]
    kw = dict(dtype=torch.float64, device="cuda")
    tau = torch.empty(B, T, nt, **kw); lam = torch.empty(B, 2 * T * m, **kw); slack = torch.empty(B, 2 * T * m, **kw)
    nu = torch.empty(B, T * n, **kw); info = torch.empty(B, 2, dtype=torch.int32, device="cuda")
    ws = torch.empty(int(lib.dqp_mpc_qp_workspace_bytes(ctypes.byref(dims))) // 8, **kw)
    P = lambda x: ctypes.c_void_p(x.data_ptr())
    resid = torch.empty(B, **kw)
    tb = int(lib.dqp_mpc_qp_termination_bytes(ctypes.byref(dims), ctypes.byref(opts)))
    term = torch.empty(tb // 8 + 1, **kw) if tb else None
    rc = lib.dqp_mpc_qp_forward(ctypes.byref(dims), ctypes.byref(opts), *[P(x) for x in t], P(tau), P(lam), P(nu),
                                P(slack), P(info), P(resid), P(ws), P(term) if tb else None, None)
    assert rc == 0
    torch.cuda.synchronize()
    assert int(info[:, 0].abs().max()) == 0
    assert float(resid.max()) < 1e-8, "not converged: %s" % resid.topk(4).values.tolist()
    Ct, ct, Ft, ft, x0t = t[:5]
    tk = tau.transpose(0, 1)                                            # (T, B, nt)
    x, u = tk[..., :n], tk[..., n:]
    scale = float(tk.abs().max())
    # feasibility: dynamics, initial state, bounds (with the slacks)
    dyn = (Ft @ tk[:-1].unsqueeze(-1)).squeeze(-1) + ft - x[1:]
    assert float(dyn.abs().max()) < 1e-8 * max(1.0, scale)
    assert float((x[0] - x0t).abs().max()) < 1e-9
    lu, ll = lam[:, :T * m].reshape(B, T, m).transpose(0, 1), lam[:, T * m:].reshape(B, T, m).transpose(0, 1)
    su, sl = slack[:, :T * m].reshape(B, T, m).transpose(0, 1), slack[:, T * m:].reshape(B, T, m).transpose(0, 1)
    assert float((u - dev(hi) + su).abs().max()) < 1e-8 and float((-u + dev(lo) + sl).abs().max()) < 1e-8
    assert float(lam.min()) > 0 and float(slack.min()) > 0
    assert float((lam * slack).max()) < 1e-8
    # stationarity per knot: C tau + c + [0; lu - ll] + F' nu_t - [nu_{t-1}; 0] (+ [nu_init; 0] at t = 0)
    nuk = nu.reshape(B, T, n).transpose(0, 1)
    g = (Ct @ tk.unsqueeze(-1)).squeeze(-1) + ct
    g[..., n:] += lu - ll
    g[:-1] += (Ft.transpose(-1, -2) @ nuk[:-1].unsqueeze(-1)).squeeze(-1)
    g[1:, :, :n] -= nuk[:-1]
    g[0, :, :n] += nuk[-1]
    assert float(g.abs().max()) < 1e-7 * max(1.0, float(nu.abs().max()))


@pytest.mark.parametrize("name,T", [("RIC_n12_m4_T6_b4", 6), ("RIC_n12_m4_T10_b3", 10)])
@pytest.mark.parametrize("tag,kw", [("single", dict(single_qp_solve=True)), ("sqp", dict(qp_iter=3))])
def test_mpc_mirror_vs_reference_n12_m4(name, T, tag, kw):
    """qp_wrapper.MPC end to end at n_state 12, n_ctrl 4 (the QP is nz = 96 / 160: beyond every dense
    kernel, served by the stage-wise kernels) against the reference's own qp_wrapper.MPC outputs
    (tests/golden/make_golden_ric.py): x, u rtol 1e-6 / atol 1e-8, gradients rtol 1e-4 / atol 1e-6."""
    from diff_qp_mpc_amd.qp_wrapper import MPC, QuadCost, LinDx
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    B, n, m = g["in_x0"].shape[0], 12, 4
    C, c, F, f, x0 = [dev(g["in_" + k], grad=True) for k in ("C", "c", "F", "f", "x0")]
    mpc = MPC(n, m, T, u_lower=dev(g["in_u_lower"]), u_upper=dev(g["in_u_upper"]), n_batch=B, verbose=-1, **kw)
    x, u = mpc(x0, QuadCost(C, c), LinDx(F, f), None)
    np.testing.assert_allclose(x.detach().cpu().numpy(), g["%s_x" % tag], **ZT)
    np.testing.assert_allclose(u.detach().cpu().numpy(), g["%s_u" % tag], **ZT)
    (x.sum() + 2.0 * u.sum()).backward()
    for k, t in (("C", C), ("c", c), ("F", F), ("f", f), ("x0", x0)):
        got = t.grad.cpu().numpy() if t.grad is not None else np.zeros(t.shape)
        np.testing.assert_allclose(got, g["%s_d%s" % (tag, k)], err_msg="%s d%s" % (tag, k), **GT)


@pytest.mark.parametrize("robot,T", [("cartpole1l", 5), ("pendulum_dx", 10), ("cartpole2l", 4)])
def test_true_dynamics_residual_stagewise_equals_dense(robot, T):
    """qp_wrapper.MPC on a registered nonlinear model: the equality residual of the PDIPM iterations is
    the model's true step (the reference's dyn_res closure, qp_wrapper.py:309,316).  The stage-wise
    kernels evaluate it per knot; the dense one-QP-per-wavefront kernels do the same through
    dqp_opts.dyn_* (pinned by the reference's config-2 golden).  Same trajectories and gradients."""
    from diff_qp_mpc_amd import qp_wrapper
    from diff_qp_mpc_amd.dynamics import DeviceDynamics
    dyn = DeviceDynamics(robot)
    n, m, B = dyn.n_state, dyn.n_ctrl, 12
    gen = torch.Generator().manual_seed(T)
    rnd = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float64).cuda()
    x0 = 0.3 * rnd(B, n)
    if robot == "pendulum_dx":
        x0[:, :2] = torch.nn.functional.normalize(x0[:, :2] + torch.tensor([1.0, 0.0]).cuda(), dim=1)
    L = 0.3 * rnd(T, B, n + m, n + m)
    outs = {}
    for fused in (True, False):
        qp_wrapper.FUSED_MPC_QP = fused
        try:
            C = (L @ L.transpose(2, 3) + torch.eye(n + m, dtype=torch.float64, device="cuda")).requires_grad_()
            c = (0.2 * torch.ones(T, B, n + m, dtype=torch.float64, device="cuda")).requires_grad_()
            mpc = qp_wrapper.MPC(n, m, T, u_lower=-torch.ones(m).double().cuda(), u_upper=torch.ones(m).double().cuda(),
                                 n_batch=B, verbose=-1, single_qp_solve=True)
            x, u = mpc(x0, qp_wrapper.QuadCost(C, c), dyn, dyn.jac)
            (x.sum() + 2.0 * u.sum()).backward()
            outs[fused] = [t.detach().cpu().numpy() for t in (x, u, C.grad, c.grad)]
        finally:
            qp_wrapper.FUSED_MPC_QP = True
    for a, b, k in zip(outs[True], outs[False], ("x", "u", "dC", "dc")):
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-7, err_msg=k)


def test_quadrotor_interior_point_mpc_runs_at_config4_horizon():
    """qp_wrapper.MPC (interior point) on the quadrotor device model at T = 30 (nz = 480): linearisation from
    the registry's Jacobians, stage-wise PDIPM with the true RK4 step as equality residual, fused line
    search on the model.  B = 64; properties: finite, x_0 = x0, controls inside the bounds (the returned
    point is the damped QP step x + alpha dx of qp_wrapper.py:298-324, not a rollout), finite gradients."""
    from diff_qp_mpc_amd import qp_wrapper
    from diff_qp_mpc_amd.dynamics import DeviceDynamics
    dyn = DeviceDynamics("rexquadrotor")
    n, m, T, B = 12, 4, 30, 64
    rng = np.random.default_rng(0)
    x0 = dev(rng.uniform(-1, 1, (B, n)) * np.array([1.0] * 3 + [0.15] * 3 + [0.5] * 3 + [0.25] * 3))
    Qw = torch.tensor([10.0] * 3 + [0.01] * 3 + [1.0] * 3 + [0.01] * 3 + [1e-4] * m, dtype=torch.float64, device="cuda")
    hover = (2.0 * 9.81 + 4 * 30.48576) / (4 * 0.0244101 * 100.0)
    C = torch.diag(Qw).repeat(T, B, 1, 1).requires_grad_()
    ref = torch.cat([torch.zeros(n, dtype=torch.float64, device="cuda"), torch.full((m,), hover, dtype=torch.float64, device="cuda")])
    c = (-(Qw * ref)).repeat(T, B, 1).requires_grad_()
    lo, hi = torch.full((m,), 11.5, dtype=torch.float64, device="cuda"), torch.full((m,), 18.3, dtype=torch.float64, device="cuda")
    mpc = qp_wrapper.MPC(n, m, T, u_lower=lo, u_upper=hi, n_batch=B, verbose=-1, qp_iter=2,
                         u_init=torch.full((T, B, m), hover, dtype=torch.float64, device="cuda"))
    x, u = mpc(x0, qp_wrapper.QuadCost(C, c), dyn, dyn.jac)
    assert x.shape == (T, B, n) and u.shape == (T, B, m)
    assert bool(torch.isfinite(x).all()) and bool(torch.isfinite(u).all())
    assert float((x[0] - x0).detach().abs().max()) < 1e-9
    assert float(u.min()) >= 11.5 - 1e-6 and float(u.max()) <= 18.3 + 1e-6
    (x.sum() + u.sum()).backward()
    assert bool(torch.isfinite(C.grad).all()) and bool(torch.isfinite(c.grad).all())


@pytest.mark.parametrize("n,m,T,B", [(3, 3, 2, 1), (2, 1, 2, 3), (4, 2, 3, 130)])
def test_stagewise_edge_sizes(n, m, T, B):
    """Shortest horizons (T = 2: one dynamics row block), a single problem, a batch that is not a multiple
    of the four problems a wavefront holds: against the CPU oracle on the assembled QP."""
    data = problem(n, m, T, B, seed=11 * n + T + B)
    tau, grads, w = run_fused(n, m, T, data)
    Q, p, G, h, A, b = assemble(*data)
    o = oracle.dense_forward(Q, p, G, h, A, b)
    np.testing.assert_allclose(tau.reshape(B, -1), o["zhat"], **ZT)


def test_stagewise_batch_rule_equals_oracle_on_mpc_batch():
    """The batch-coupled stop on an MPC batch that triggers it early (I* < max_iter, most problems
    flagged): the stage-wise kernels' finish pass (copy of the right snapshot) against the CPU oracle,
    which runs the reference's rule literally; and the per-problem mode against the same within the
    float tolerance."""
    from diff_qp_mpc_amd import qp as qpmod
    n, m, T, B = 4, 2, 6, 64
    data = problem(n, m, T, B, seed=99, active=0.2)
    Q, p, G, h, A, b = assemble(*data)
    o = oracle.dense_forward(Q, p, G, h, A, b)
    assert o["iters"] < 20                                     # the rule fired before max_iter
    tau_b, _, _ = run_fused(n, m, T, data)
    np.testing.assert_allclose(tau_b.reshape(B, -1), o["zhat"], **ZT)
    old = qpmod.TERMINATION
    qpmod.TERMINATION = "per_problem"
    try:
        tau_p, _, _ = run_fused(n, m, T, data)
    finally:
        qpmod.TERMINATION = old
    np.testing.assert_allclose(tau_p.reshape(B, -1), o["zhat"], rtol=1e-5, atol=1e-7)


def test_riccati_kkt_solve_against_a_dense_solve():
    """One KKT solve of the stage-wise kernels (dqp_mpc_qp_backward: Riccati factorisation with d = lam/slack,
    right-hand side (g, 0, 0, 0)) at an arbitrary interior point against numpy's dense solve of the assembled
    KKT matrix: dc = dx, df_t = dnu_t, dx0 = -dnu_init (n 12, m 4, T 8: a 408 x 408 system)."""
    from diff_qp_mpc_amd import _lib
    n, m, T, B = 12, 4, 8, 4
    nt = n + m
    C, c, F, f, x0, lo, hi = problem(n, m, T, B, seed=21)
    Q, p, G, h, A, b = assemble(C, c, F, f, x0, lo, hi)
    rng = np.random.default_rng(5)
    nz, nineq, neq = T * nt, 2 * T * m, T * n
    s = rng.random((B, nineq)) + 0.1; z = rng.random((B, nineq)) + 0.1
    tau = rng.standard_normal((B, T, nt)); nu = rng.standard_normal((B, neq)); g = rng.standard_normal((B, T, nt))
    dx_ref = np.zeros((B, nz)); dy_ref = np.zeros((B, neq))
    for i in range(B):
        D = np.diag(z[i] / s[i])
        Z = np.zeros
        K = np.block([[Q[i], Z((nz, nineq)), G[i].T, A[i].T],
                      [Z((nineq, nz)), D, np.eye(nineq), Z((nineq, neq))],
                      [G[i], np.eye(nineq), Z((nineq, nineq)), Z((nineq, neq))],
                      [A[i], Z((neq, nineq)), Z((neq, nineq)), Z((neq, neq))]])
        sol = np.linalg.solve(K, -np.concatenate([g[i].reshape(-1), np.zeros(2 * nineq + neq)]))
        dx_ref[i] = sol[:nz]; dy_ref[i] = sol[nz + 2 * nineq:]
    lib = _lib.load()
    dims = _lib.dqp_mpc_dims(B, n, m, T, 1, 0)
    opts = _lib.dqp_opts(0.0, 0.0, 0, 0, _lib.DQP_FLAG_DENSE_BACKWARD, 0)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda")
    Ct, Ft, taut, lamt, nut, st, gt = t(C), t(F), t(tau), t(z), t(nu), t(s), t(g)
    kw = dict(dtype=torch.float64, device="cuda")
    dC, dc = torch.empty(T, B, nt, nt, **kw), torch.empty(T, B, nt, **kw)
    dF, df, dx0 = torch.empty(T - 1, B, n, nt, **kw), torch.empty(T - 1, B, n, **kw), torch.empty(B, n, **kw)
    ws = torch.empty(int(lib.dqp_mpc_qp_workspace_bytes(ctypes.byref(dims))) // 8, **kw)
    P = lambda x: ctypes.c_void_p(x.data_ptr())
    rc = lib.dqp_mpc_qp_backward(ctypes.byref(dims), ctypes.byref(opts), P(Ct), P(Ft), P(taut), P(lamt), P(nut), P(st), P(gt),
                                 P(dC), P(dc), P(dF), P(df), P(dx0), None, P(ws), None)
    assert rc == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(dc.cpu().numpy().transpose(1, 0, 2).reshape(B, -1), dx_ref, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(df.cpu().numpy().transpose(1, 0, 2).reshape(B, -1), dy_ref[:, :(T - 1) * n], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(dx0.cpu().numpy(), -dy_ref[:, (T - 1) * n:], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("n,m,T,B", [(5, 1, 6, 7), (8, 2, 5, 5), (6, 3, 4, 6), (10, 4, 3, 4), (2, 2, 5, 9)])
def test_stagewise_more_size_pairs(n, m, T, B):
    """Other compiled (n_state, n_ctrl) pairs of the stage-wise kernels against the CPU oracle."""
    data = problem(n, m, T, B, seed=5 * n + m + T)
    tau, grads, w = run_fused(n, m, T, data)
    Q, p, G, h, A, b = assemble(*data)
    o = oracle.dense_forward(Q, p, G, h, A, b)
    np.testing.assert_allclose(tau.reshape(B, -1), o["zhat"], **ZT)
    og = oracle.dense_backward(o["K"], o["zhat"], o["lam"], o["nu"], w.reshape(B, -1))
    nt = n + m
    dc = np.stack([og["dp"][:, t * nt:(t + 1) * nt] for t in range(T)])
    np.testing.assert_allclose(grads[1], dc, err_msg="dc", **GT)


ALL_PAIRS = [(12, 4), (3, 3), (3, 1), (4, 1), (6, 1), (2, 1), (4, 2), (5, 1), (8, 1), (2, 2), (3, 2), (6, 2), (8, 2), (6, 3),
             (4, 4), (8, 4), (10, 4), (12, 2)]        # DQP_RIC_SIZES of csrc/dqp_ric.hip


@pytest.mark.parametrize("n,m,T,B", [(3, 1, 10, 9), (4, 2, 6, 130), (12, 4, 4, 6)] + [(n, m, 5, 6) for n, m in ALL_PAIRS])
def test_stagewise_lds_resident_equals_global_workspace(n, m, T, B):
    """Short horizons keep the whole problem (C, F, c, f, iterates, factors) in LDS; DQP_FLAG_RIC_GLOBAL_WS pins the
    same sizes to the caller's workspace (the path long horizons take, LDS-DMA prefetch of every knot).  Same
    arithmetic in the same order: identical outputs, and both against the CPU oracle."""
    from diff_qp_mpc_amd import _lib, qp_wrapper
    data = problem(n, m, T, B, seed=5 * n + T)
    tau_l, grads_l, w = run_fused(n, m, T, data)
    old = qp_wrapper.EXTRA_FLAGS
    qp_wrapper.EXTRA_FLAGS = _lib.DQP_FLAG_RIC_GLOBAL_WS
    try:
        tau_g, grads_g, _ = run_fused(n, m, T, data)
    finally:
        qp_wrapper.EXTRA_FLAGS = old
    np.testing.assert_array_equal(tau_l, tau_g)
    for a, b in zip(grads_l, grads_g):
        np.testing.assert_array_equal(a, b)
    Q, p, G, h, A, b = assemble(*data)
    o = oracle.dense_forward(Q, p, G, h, A, b)
    np.testing.assert_allclose(tau_l.reshape(B, -1), o["zhat"], **ZT)


@pytest.mark.parametrize("n,m,T,B", [(12, 4, 30, 37), (4, 2, 6, 9)])
def test_stagewise_inputs_and_workspace_only_8_byte_aligned(n, m, T, B):
    """The knot prefetch moves 16-byte pieces by LDS-DMA; a caller's C, c, F, f and workspace are doubles and
    promise 8-byte alignment only (a view into a larger tensor): same result bit for bit."""
    from diff_qp_mpc_amd import _lib
    lib = _lib.load()
    C0, c0, F0, f0, x0, lo, hi = [dev(a) for a in problem(n, m, T, B, seed=3 * n + T)]
    nt = n + m
    kw = dict(dtype=torch.float64, device="cuda")
    dims = _lib.dqp_mpc_dims(B, n, m, T, 1, 0)
    wsb = int(lib.dqp_mpc_qp_workspace_bytes(ctypes.byref(dims)))
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    outs = []
    for shift in (0, 1):
        def placed(t):
            buf = torch.empty(t.numel() + 2, **kw)
            v = buf[shift:shift + t.numel()]
            v.copy_(t.reshape(-1))
            return v
        C, c, F, f = placed(C0), placed(c0), placed(F0), placed(f0)
        ws = torch.empty(wsb // 8 + 2, **kw)[shift:]
        assert C.data_ptr() % 16 == 8 * shift and ws.data_ptr() % 16 == 8 * shift
        tau = torch.empty(B, T, nt, **kw); lam = torch.empty(B, 2 * T * m, **kw); slack = torch.empty(B, 2 * T * m, **kw)
        nu = torch.empty(B, T * n, **kw); info = torch.empty(B, 2, dtype=torch.int32, device="cuda"); resid = torch.empty(B, **kw)
        opts = _lib.dqp_opts(1e-12, 1e-10, 20, 3, _lib.DQP_FLAG_RIC_GLOBAL_WS, 0)
        rc = lib.dqp_mpc_qp_forward(ctypes.byref(dims), ctypes.byref(opts), P(C), P(c), P(F), P(f), P(x0), P(lo), P(hi),
                                    P(tau), P(lam), P(nu), P(slack), P(info), P(resid), P(ws), None, None)
        torch.cuda.synchronize()
        assert rc == 0
        assert float(resid.max()) < 1e-9
        outs.append(tau.cpu().numpy())
    np.testing.assert_array_equal(outs[0], outs[1])
