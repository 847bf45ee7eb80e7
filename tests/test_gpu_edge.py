"""Degenerate inputs on the GPU (the assertions behind tools/robustness_probe.py), all three kernel
families: rank-deficient A, non-SPD Q, infeasible constraints, NaN inputs, badly scaled costs and
fp32 inputs.  Where the reference's behaviour is defined it is the yardstick (the oracle on the
same inputs); where the reference raises or returns garbage the contract of include/dqp.h holds:
per-problem status, finite work, neighbours in the batch unaffected."""
import numpy as np
import pytest
import torch

from oracle import oracle
from families import family

pytestmark = pytest.mark.gpu
B, NZ, NINEQ, NEQ = 64, 30, 30, 15


@pytest.fixture(params=["auto", "rows", "generic"])
def fam(request):
    from diff_qp_mpc_amd import qp as qpmod, _lib
    qpmod.FORCE_FLAGS = {"auto": 0, "rows": _lib.DQP_FLAG_NO_NULLSPACE,
                         "generic": _lib.DQP_FLAG_GENERIC_ONLY}[request.param]
    yield request.param
    qpmod.FORCE_FLAGS = 0


def dev(a, dtype=torch.float64):
    return torch.tensor(np.asarray(a), dtype=dtype, device="cuda")


def base():
    return [a.copy() for a in family(5, B, NZ, NINEQ, NEQ, "R")]


def solve(ins, termination=None):
    from diff_qp_mpc_amd import qp as qpmod
    out = qpmod._forward_impl(*[dev(a) for a in ins], 1e-12, 20, 3, termination=termination)
    torch.cuda.synchronize()
    return out


def test_rank_deficient_A_is_flagged(fam):
    """Two identical equality rows: A Q^-1 A^T is singular.  The reference's pre_factor_kkt breaks
    on it (the oracle returns its LU failure); here EVERY such problem reports
    DQP_STATUS_A_RANK_DEF, the launch finishes, and the other problems of the batch are untouched."""
    from diff_qp_mpc_amd import _lib
    Q, p, G, h, A, b = base()
    bad = np.arange(0, B, 3)
    A[bad, 1] = A[bad, 0]; b[bad, 1] = b[bad, 0]
    zhat, lam, nu, slack, info, resid, _ = solve([Q, p, G, h, A, b])
    st = info[:, 0].cpu().numpy()
    assert (st[bad] == _lib.DQP_STATUS_A_RANK_DEF).all(), st[bad]
    good = np.setdiff1d(np.arange(B), bad)
    assert (st[good] == 0).all()
    clean = solve(base())
    np.testing.assert_allclose(zhat.cpu().numpy()[good], clean[0].cpu().numpy()[good], rtol=1e-9, atol=1e-11)


def test_non_spd_Q_is_flagged_per_problem(fam):
    from diff_qp_mpc_amd import _lib
    import diff_qp_mpc_amd as dqp
    Q, p, G, h, A, b = base()
    Q[7] = -Q[7]
    zhat, lam, nu, slack, info, resid, _ = solve([Q, p, G, h, A, b])
    st = info[:, 0].cpu().numpy()
    assert st[7] == _lib.DQP_STATUS_Q_NOT_PD and (np.delete(st, 7) == 0).all()
    assert bool(torch.isfinite(zhat[torch.arange(B) != 7]).all())
    from diff_qp_mpc_amd import qp as qpmod
    with pytest.raises(RuntimeError, match="Q is not SPD"):                       # qp.py:86
        # lazy by default: forward enqueues the check (no synchronisation), flush_checks() / backward / a later
        # forward raises
        dqp.QPFunction(check_Q_spd=True, verbose=-1)(*[dev(a) for a in (Q, p, G, h, A, b)])
        qpmod.flush_checks()
    assert qpmod._pending == []
    z = dqp.QPFunction(check_Q_spd=True, verbose=-1)(*[dev(a).requires_grad_() for a in (Q, p, G, h, A, b)])
    with pytest.raises(RuntimeError, match="Q is not SPD"):                       # ... at backward
        z.sum().backward()
    qpmod.CHECKS = "sync"                                                         # as the reference: inside forward
    try:
        with pytest.raises(RuntimeError, match="Q is not SPD"):
            dqp.QPFunction(check_Q_spd=True, verbose=-1)(*[dev(a) for a in (Q, p, G, h, A, b)])
    finally:
        qpmod.CHECKS = "lazy"
    assert qpmod._pending == []


def test_infeasible_problem_prints_warning_and_returns_best_iterate(fam, capsys):
    """G z <= h with two contradicting rows: no feasible point.  The reference prints INACC_ERR when
    the best residual stays above 1 (batch.py:142-143) and returns its best iterate; same here, and
    the iterate is the oracle's to the usual tolerance on the feasible problems of the batch."""
    import diff_qp_mpc_amd as dqp
    Q, p, G, h, A, b = base()
    G[0, 1] = -G[0, 0]; h[0, 0] = -1.0; h[0, 1] = -1.0
    o = oracle.qp_forward(Q, p, G, h, A, b)
    assert o["best_resid"][0] > 1.0
    from diff_qp_mpc_amd import qp as qpmod
    zhat = dqp.QPFunction(check_Q_spd=False, verbose=0)(*[dev(a) for a in (Q, p, G, h, A, b)])
    qpmod.flush_checks()                                    # the warning is lazy too (qp.CHECKS)
    assert "Returning an inaccurate and potentially incorrect solution" in capsys.readouterr().out
    assert bool(torch.isfinite(zhat).all())
    ok = o["best_resid"] < 1e-8
    assert not ok[0] and ok.mean() > 0.9
    np.testing.assert_allclose(zhat.cpu().numpy()[ok], o["zhat"][ok], rtol=1e-6, atol=1e-8)
    # silent with verbose = -1, as the reference
    dqp.QPFunction(check_Q_spd=False, verbose=-1)(*[dev(a) for a in (Q, p, G, h, A, b)])
    qpmod.flush_checks()
    assert capsys.readouterr().out == ""


def test_nan_input_stays_in_its_problem(fam):
    Q, p, G, h, A, b = base()
    p[3, 2] = np.nan
    keep = np.arange(B) != 3
    # per-problem termination: problems never interact -> bitwise identical neighbours
    zhat, lam, nu, slack, info, resid, saved = solve([Q, p, G, h, A, b], "per_problem")
    z = zhat.cpu().numpy()
    assert not np.isfinite(z[3]).all()
    np.testing.assert_array_equal(z[keep], solve(base(), "per_problem")[0].cpu().numpy()[keep])
    assert int(info[3, 1]) <= 20
    # batch termination (the reference's rule): the NaN sample never improves and never converges, so
    # it only shifts where the batch stops -- neighbours agree to round-off, as in the reference
    z = solve([Q, p, G, h, A, b], "batch")[0].cpu().numpy()
    assert not np.isfinite(z[3]).all()
    np.testing.assert_allclose(z[keep], solve(base(), "batch")[0].cpu().numpy()[keep], rtol=1e-9, atol=1e-11)


def test_badly_scaled_cost_matches_the_reference_behaviour(fam):
    """Q, p scaled by 1e6: the reference's PDIPM (no scaling of its own) breaks down after five
    iterations with residuals ~1e3; the kernels follow the same trajectory and stop with it."""
    Q, p, G, h, A, b = base()
    Q *= 1e6; p *= 1e6
    o = oracle.qp_forward(Q, p, G, h, A, b)
    zhat, lam, nu, slack, info, resid, _ = solve([Q, p, G, h, A, b])
    assert bool(torch.isfinite(zhat).all())
    r = resid.cpu().numpy()
    assert np.median(r) > 1.0 and np.median(o["best_resid"]) > 1.0
    np.testing.assert_allclose(np.log10(r), np.log10(o["best_resid"]), atol=0.5)
    Q, p, G, h, A, b = base()
    Q *= 1e-6; p *= 1e-6                                             # the benign direction converges
    o = oracle.qp_forward(Q, p, G, h, A, b)
    zhat = solve([Q, p, G, h, A, b])[0]
    cm = o["best_resid"] < 1e-8
    assert cm.mean() > 0.95
    np.testing.assert_allclose(zhat.cpu().numpy()[cm], o["zhat"][cm], rtol=1e-6, atol=1e-8)


def test_fp32_inputs_round_trip(fam):
    """The reference's profilers run fp32 (prof-linear.py:64-75).  The kernels compute in fp64; fp32
    tensors are up-cast on the way in and results / gradients come back as fp32.  Against the
    oracle on the fp32-rounded inputs: zhat rtol 1e-4 / atol 1e-5, gradients rtol 1e-3 / atol 1e-4."""
    import diff_qp_mpc_amd as dqp
    ins64 = base()
    ins32 = [dev(a, torch.float32).requires_grad_() for a in ins64]
    zhat = dqp.QPFunction(check_Q_spd=False, verbose=-1)(*ins32)
    assert zhat.dtype == torch.float32
    rounded = [t.detach().double().cpu().numpy() for t in ins32]
    o = oracle.qp_forward(*rounded)
    cm = o["best_resid"] < 1e-8
    np.testing.assert_allclose(zhat.detach().cpu().numpy()[cm], o["zhat"][cm], rtol=1e-4, atol=1e-5)
    zhat.sum().backward()
    og = oracle.qp_backward(rounded[0], rounded[2], rounded[4], o["zhat"], o["lam"], o["nu"], o["slack"],
                            np.ones((B, NZ)))
    gm = cm & (np.maximum(o["lam"], o["slack"]).min(1) > 1e-5)
    for k, t in zip("QpGhAb", ins32):
        assert t.grad.dtype == torch.float32
        np.testing.assert_allclose(t.grad.cpu().numpy()[gm], og["d" + k][gm], rtol=1e-3, atol=1e-4, err_msg="d" + k)


def test_sl1qp_reformulation():
    """sl1qp.sl1qpify (the formulation of sl1qp_mpc.py:703-752 at general sizes, neq != nineq): the
    extended QP solved on the GPU equals the CPU oracle on the same extended QP; for mu above the
    multipliers the exact penalty returns the original QP's solution; for a small mu on an
    over-constrained problem the constraints are softened (violations allowed, finite solution)."""
    import diff_qp_mpc_amd as dqp
    from diff_qp_mpc_amd import sl1qp
    B, nz, nineq, neq = 7, 10, 8, 4
    Q, p, G, h, A, b = [dev(a).requires_grad_() for a in family(0, B, nz, nineq, neq)]
    z_hard = dqp.DenseQPFunction(verbose=-1)(Q, p, G, h, A, b)
    ext = sl1qp.sl1qpify(Q, p, G, h, A, b, mu=200.0)
    assert ext[0].shape == (B, nz + 2 * neq + nineq, nz + 2 * neq + nineq)
    assert ext[2].shape == (B, 2 * nineq + 2 * neq, nz + 2 * neq + nineq) and ext[4].shape == (B, neq, nz + 2 * neq + nineq)
    z_soft = sl1qp.SL1QPFunction(mu=200.0, verbose=-1)(Q, p, G, h, A, b)
    o = oracle.dense_forward(*[t.detach().cpu().numpy() for t in ext])
    np.testing.assert_allclose(z_soft.detach().cpu().numpy(), o["zhat"][:, :nz], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(z_soft.detach().cpu().numpy(), z_hard.detach().cpu().numpy(), rtol=1e-3, atol=1e-4)
    z_soft.sum().backward()
    assert all(t.grad is not None and bool(torch.isfinite(t.grad).all()) for t in (Q, p, G, h, A, b))
    # infeasible hard problem (two contradictory equality rows): the l1 form still has a solution
    A2 = torch.cat([A.detach(), A.detach()[:, :1]], 1)
    b2 = torch.cat([b.detach(), b.detach()[:, :1] + 1.0], 1)
    z = sl1qp.SL1QPFunction(mu=5.0, verbose=-1)(Q.detach(), p.detach(), G.detach(), h.detach(), A2, b2)
    assert bool(torch.isfinite(z).all())
    gap = (A2 @ z.unsqueeze(-1)).squeeze(-1) - b2
    assert float(gap[:, [0, neq]].abs().sum(1).min()) > 0.5          # the contradiction is absorbed by the slacks
