"""GPU parity tests (run on a real MI355X with `-m gpu`): the HIP path, called through the
C ABI, against (1) the golden vectors captured from the reference and (2) the CPU oracle on
seeded inputs, plus size-independent KKT properties at the full benchmark size.

Tolerances (fp64), stated per SURVEY.md §8c: the reference's own two solvers agree to 3e-13 on
zhat and 5e-6 on gradients.  The default termination is the reference's batch-coupled rule replayed
on the device (every test also runs the opt-in per-problem mode where it says so); the kernels work
in different coordinates than the reference, so parity is a float tolerance, never bit-exact:
    zhat            rtol 1e-6  atol 1e-8
    lam, nu, slack  rtol 1e-5  atol 1e-7
    gradients       rtol 1e-4  atol 1e-6
"""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import oracle
from families import family, family_mpc

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "[RM]_*.npz")))
DENSE = [c for c in CASES if "dense_zhat" in np.load(os.path.join(GOLDEN, c + ".npz")).files]

ZT = dict(rtol=1e-6, atol=1e-8)
DT = dict(rtol=1e-5, atol=1e-7)
GT = dict(rtol=1e-4, atol=1e-6)


@pytest.fixture(params=["auto", "rows", "generic"], autouse=True)
def kernel_family(request):
    """Every test runs against all three kernel families: automatic dispatch (DPP-row kernels
    where instantiated, null-space forward), the DPP-row forward that keeps the equality rows
    (what runs when the caller passes no workspace), and the generic one-QP-per-wavefront LDS
    kernels."""
    from diff_qp_mpc_amd import qp as qpmod, _lib
    qpmod.FORCE_FLAGS = {"auto": 0, "rows": _lib.DQP_FLAG_NO_NULLSPACE,
                         "generic": _lib.DQP_FLAG_GENERIC_ONLY}[request.param]
    yield request.param
    qpmod.FORCE_FLAGS = 0


@pytest.fixture(scope="module")
def dqp():
    assert torch.cuda.is_available(), "these tests need a GPU"
    import diff_qp_mpc_amd
    from diff_qp_mpc_amd import _lib
    _lib.load()          # fails loudly if the HIP library is missing
    return diff_qp_mpc_amd


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def dev(a, grad=True):
    t = torch.tensor(np.asarray(a), dtype=torch.float64, device="cuda")
    return t.requires_grad_() if grad else t


def family_R(seed, B, nz, nineq, neq):
    return family(seed, B, nz, nineq, neq, "R")


def over_tolerance(a, b):
    """per-sample: do the forward outputs of two runs differ beyond the test tolerances (NaN counts as differing)?"""
    bad = np.zeros(a["zhat"].shape[0], dtype=bool)
    for k, tol in (("zhat", ZT), ("lam", DT), ("nu", DT), ("slack", DT)):
        e = np.abs(a[k] - b[k]) > tol["atol"] + tol["rtol"] * np.abs(b[k])
        bad |= (e | np.isnan(a[k]) | np.isnan(b[k])).reshape(e.shape[0], -1).any(1)
    return bad


def reference_outputs(ins, ct=None):
    """Oracle forward (and, with a cotangent, backward) for a stress batch -> (outputs, differ[, gradients]).
    Expected = the LITERAL restatement of batch.py (unguarded get_step, batch.py:211-214) on every sample, except
    those where the literal run and the guarded run (the reference's own batch_LU.get_step, batch_LU.py:203-214)
    actually DIFFER beyond the test tolerances, in an output or in a gradient: there the literal reference froze its
    iterate on a division by an exactly-zero step component one or two iterations short of convergence (which moves
    d = lam/slack of weakly active constraints, hence the gradients) -- an accident of its arithmetic order that
    kernels working in other coordinates do not share -- and the guarded run is the expectation.  Samples whose
    literal history turns NaN only after convergence (most of them: the two runs then agree) stay with the literal
    oracle.  Both variants are pinned by tests/golden (test_oracle_golden.py).  Counts over 8192 samples per family:
    tools/strict_vs_guard.py, profiles/r3/strict_vs_guard.log."""
    lit = oracle.qp_forward(*ins)
    grd = oracle.qp_forward(*ins, guard=True)
    differ = over_tolerance(lit, grd)
    gl = gg = None
    if ct is not None:
        gl = oracle.qp_backward(ins[0], ins[2], ins[4], lit["zhat"], lit["lam"], lit["nu"], lit["slack"], ct)
        gg = oracle.qp_backward(ins[0], ins[2], ins[4], grd["zhat"], grd["lam"], grd["nu"], grd["slack"], ct)
        for k in gl:
            e = ~(np.abs(gl[k] - gg[k]) <= GT["atol"] + GT["rtol"] * np.abs(gg[k]))
            differ |= e.reshape(e.shape[0], -1).any(1)
        for k in gl:
            gl[k][differ] = gg[k][differ]
    for k in ("zhat", "lam", "nu", "slack", "best_resid"):
        lit[k][differ] = grd[k][differ]
    return (lit, differ) if ct is None else (lit, differ, gl)


@pytest.mark.parametrize("name", CASES)
def test_forward_backward_vs_golden(dqp, name):
    g = load(name)
    neq = g["nu"].shape[1]
    ins = [dev(g["in_" + k]) for k in "QpGhAb"]
    for tag in ("ones", "rand"):
        for t in ins:
            t.grad = None
        zhat = dqp.QPFunction(check_Q_spd=True, verbose=-1)(*ins)
        np.testing.assert_allclose(zhat.detach().cpu().numpy(), g["zhat"], **ZT)
        zhat.backward(dev(g["ct_" + tag], grad=False))
        for k, t in zip("QpGhAb", ins):
            if neq == 0 and k in "Ab":
                continue
            np.testing.assert_allclose(t.grad.cpu().numpy(), g["d%s_%s" % (k, tag)],
                                       err_msg="d%s (%s)" % (k, tag), **GT)


@pytest.mark.parametrize("name", CASES)
def test_duals_vs_golden_through_c_abi(dqp, name):
    """lam / nu / slack straight from the C ABI entry point."""
    from diff_qp_mpc_amd import qp as qpmod
    g = load(name)
    ins = [dev(g["in_" + k], grad=False) for k in "QpGhAb"]
    zhat, lam, nu, slack, info, resid, _ = qpmod._forward_impl(*ins, 1e-12, 20, 3)
    assert int(info[:, 0].abs().max()) == 0
    np.testing.assert_allclose(zhat.cpu().numpy(), g["zhat"], **ZT)
    np.testing.assert_allclose(lam.cpu().numpy(), g["lam"], **DT)
    np.testing.assert_allclose(slack.cpu().numpy(), g["slack"], **DT)
    if g["nu"].shape[1]:
        np.testing.assert_allclose(nu.cpu().numpy(), g["nu"], **DT)
    assert float(resid.max()) < 1e-6


@pytest.mark.parametrize("name", DENSE)
def test_dense_vs_golden(dqp, name):
    g = load(name)
    ins = [dev(g["in_" + k]) for k in "QpGhAb"]
    for tag in ("ones", "rand"):
        for t in ins:
            t.grad = None
        zhat = dqp.DenseQPFunction()(*ins, None)
        np.testing.assert_allclose(zhat.detach().cpu().numpy(), g["dense_zhat"], **ZT)
        zhat.backward(dev(g["ct_" + tag], grad=False))
        for k, t in zip("QpGhAb", ins):
            np.testing.assert_allclose(t.grad.cpu().numpy(), g["dense_d%s_%s" % (k, tag)],
                                       err_msg="dense d%s (%s)" % (k, tag), **GT)


@pytest.mark.parametrize("shape", [(64, 30, 30, 15), (33, 40, 20, 30), (17, 7, 64, 3),
                                   (9, 64, 64, 32), (5, 1, 1, 1), (130, 12, 9, 0),
                                   (70, 20, 10, 15), (41, 15, 10, 10), (23, 25, 10, 20),
                                   (37, 35, 10, 30), (6, 10, 5, 3)])
def test_vs_oracle_seeded(dqp, shape):
    B, nz, nineq, neq = shape
    Q, p, G, h, A, b = family_R(100 + nz, B, nz, nineq, max(neq, 1))
    if neq == 0:
        A = np.zeros((B, 0, nz)); b = np.zeros((B, 0))
    o = oracle.qp_forward(Q, p, G, h, A, b)
    ins = [dev(a) for a in (Q, p, G, h, A, b)]
    zhat = dqp.QPFunction(check_Q_spd=False, verbose=-1)(*ins)
    np.testing.assert_allclose(zhat.detach().cpu().numpy(), o["zhat"], **ZT)
    ct = np.random.default_rng(0).standard_normal((B, nz))
    zhat.backward(dev(ct, grad=False))
    og = oracle.qp_backward(Q, G, A, o["zhat"], o["lam"], o["nu"], o["slack"], ct)
    for k, t in zip("QpGhAb", ins):
        if neq == 0 and k in "Ab":
            continue
        np.testing.assert_allclose(t.grad.cpu().numpy(), og["d" + k], err_msg="d" + k, **GT)


@pytest.mark.parametrize("kind,seed", [("M", 0), ("M", 1), ("D", 1), ("D", 3)])
def test_stress_families_metric_shape(dqp, kind, seed):
    """Reduced tools/stress_parity.py: B=2048 at the metric shape on the MPC-structured family M
    (box-constrained, many active and weakly active bounds) and the diagonal-cost family D, every
    output and every gradient per problem against the oracle.  These are the four batches that
    held the round-1 outliers (M seed 0 #174/#1889, M seed 1 #1206, D seed 1 #1782, D seed 3
    #1767): there the reference's batch.py get_step divides by an exactly-zero step component,
    its iterate turns NaN and its best iterate is frozen 1-2 iterations early, which moves
    d = lam/slack of weakly active constraints (parity section)."""
    B, nz, nineq, neq = 2048, 30, 30, 15
    ins_np = family_mpc(seed, B) if kind == "M" else family(1000 * seed + nz, B, nz, nineq, neq, kind)
    ct = np.random.default_rng(seed).standard_normal((B, nz))
    o, differ, og = reference_outputs(ins_np, ct)
    # literal parity on all but the samples where the reference's two step rules part ways (printed with -s)
    print("literal and guarded reference differ on", np.nonzero(differ)[0])
    assert differ.sum() <= 2, np.nonzero(differ)[0]       # M seed 0: #174 #1889; M seed 1: #1206; D seed 1: #1782; D seed 3: none
    cm = o["best_resid"] < 1e-8
    assert cm.mean() > 0.99
    from diff_qp_mpc_amd import qp as qpmod
    dv = [dev(a, grad=False) for a in ins_np]
    zhat, lam, nu, slack, info, resid, saved = qpmod._forward_impl(*dv, 1e-12, 20, 3)
    assert int(info[:, 0].abs().max()) == 0
    np.testing.assert_allclose(zhat.cpu().numpy()[cm], o["zhat"][cm], **ZT)
    np.testing.assert_allclose(lam.cpu().numpy()[cm], o["lam"][cm], **DT)
    np.testing.assert_allclose(nu.cpu().numpy()[cm], o["nu"][cm], **DT)
    np.testing.assert_allclose(slack.cpu().numpy()[cm], o["slack"][cm], **DT)
    gr = qpmod._backward_impl(saved, zhat, lam, nu, slack, dev(ct, grad=False), (True,) * 6,
                              qpmod.FORCE_FLAGS)
    # gradients are compared where strict complementarity holds (min_i max(lam_i, slack_i) > 1e-5: elsewhere
    # d = lam/slack swings by O(1) on 1e-10 changes of the iterate); that drops at most 2 of the 2048 problems
    # of these batches (1, 0, 2, 0 in the order of the parametrisation)
    gm = cm & (np.maximum(o["lam"], o["slack"]).min(1) > 1e-5)
    assert (~gm).sum() <= 2, (~gm).sum()
    for k, t in zip("QpGhAb", gr):
        np.testing.assert_allclose(t.cpu().numpy()[gm], og["d" + k][gm], err_msg="d" + k, **GT)


def test_vs_guarded_reference_golden(dqp):
    """Mz_guard_b8.npz: on samples 1 and 2 the reference as shipped freezes its iterate early
    (unguarded get_step, dQ off by up to 0.096 from its own guarded run); the kernels must agree
    with the guarded run everywhere, and with the strict run on the samples it did not affect."""
    g = load("Mz_guard_b8")
    ins = [dev(g["in_" + k]) for k in "QpGhAb"]
    zhat = dqp.QPFunction(check_Q_spd=True, verbose=-1)(*ins)
    zhat.backward(dev(g["ct"], grad=False))
    np.testing.assert_allclose(zhat.detach().cpu().numpy(), g["guard_zhat"], **ZT)
    unaffected = np.abs(g["strict_dQ"] - g["guard_dQ"]).reshape(8, -1).max(1) < 1e-9
    assert 0 < unaffected.sum() < 8
    for k, t in zip("QpGhAb", ins):
        got = t.grad.cpu().numpy()
        np.testing.assert_allclose(got, g["guard_d" + k], err_msg="d" + k, **GT)
        np.testing.assert_allclose(got[unaffected], g["strict_d" + k][unaffected], err_msg="strict d" + k, **GT)


def test_full_size_properties_and_oracle(dqp):
    """BASELINE metric config: B=4096, nz=30, nineq=30, neq=15 (family R, seed 0)."""
    B, nz, nineq, neq = 4096, 30, 30, 15
    Q, p, G, h, A, b = family_R(0, B, nz, nineq, neq)
    ins = [dev(a) for a in (Q, p, G, h, A, b)]
    from diff_qp_mpc_amd import qp as qpmod
    zhat, lam, nu, slack, info, resid, _ = qpmod._forward_impl(*ins, 1e-12, 20, 3)
    assert int(info[:, 0].abs().max()) == 0
    o = oracle.qp_forward(Q, p, G, h, A, b)
    # Problems the reference itself does not converge on in maxIter=20 (best residual stays
    # large, e.g. an oscillating iterate) are compared through their best residual only.
    conv = torch.tensor(o["best_resid"] < 1e-8, device="cuda")
    assert int((~conv).sum()) <= 4
    assert bool((resid[conv] < 1e-8).all())
    assert bool((resid[~conv] <= 2.0 * torch.tensor(o["best_resid"], device="cuda")[~conv]).all())
    Qd, pd, Gd, hd, Ad, bd = [t.detach()[conv] for t in ins]
    zc, lc, nc, sc = zhat[conv], lam[conv], nu[conv], slack[conv]
    mv = lambda M, x: torch.bmm(M, x.unsqueeze(-1)).squeeze(-1)
    mtv = lambda M, x: torch.bmm(M.transpose(1, 2), x.unsqueeze(-1)).squeeze(-1)
    stat = mv(Qd, zc) + pd + mtv(Gd, lc) + mtv(Ad, nc)
    scale = 1.0 + mv(Qd, zc).abs().max()
    assert float(stat.abs().max() / scale) < 1e-8              # stationarity
    assert float((mv(Ad, zc) - bd).abs().max()) < 1e-8         # equality feasibility
    assert float((mv(Gd, zc) + sc - hd).abs().max()) < 1e-8
    assert float(lc.min()) > 0 and float(sc.min()) > 0
    assert float((lc * sc).abs().max()) < 1e-8                 # complementarity
    # same batch through the oracle (batch-coupled termination) -> float tolerance
    cm = conv.cpu().numpy()
    np.testing.assert_allclose(zhat.cpu().numpy()[cm], o["zhat"][cm], **ZT)
    np.testing.assert_allclose(lam.cpu().numpy()[cm], o["lam"][cm], **DT)
    # backward: linearity in the cotangent + oracle
    zf = dqp.QPFunction(check_Q_spd=False, verbose=-1)(*ins)
    ct = torch.randn(B, nz, dtype=torch.float64, device="cuda",
                     generator=torch.Generator(device="cuda").manual_seed(1))
    g1 = torch.autograd.grad(zf, ins, ct, retain_graph=True)
    g2 = torch.autograd.grad(zf, ins, 2.0 * ct, retain_graph=True)
    for a, c in zip(g1, g2):
        assert torch.allclose(2.0 * a, c, rtol=1e-12, atol=1e-14)
    og = oracle.qp_backward(Q, G, A, o["zhat"], o["lam"], o["nu"], o["slack"], ct.cpu().numpy())
    for k, t in zip("QpGhAb", g1):
        np.testing.assert_allclose(t.cpu().numpy()[cm], og["d" + k][cm], err_msg="d" + k, **GT)


@pytest.mark.parametrize("B", [1, 2, 3, 5, 4099])
def test_ragged_batches_at_dpp_row_size(dqp, B):
    """Batch sizes that are not a multiple of 4 QPs per wavefront (tail rows are masked)."""
    Q, p, G, h, A, b = family_R(7, B, 30, 30, 15)
    o = oracle.qp_forward(Q, p, G, h, A, b)
    ins = [dev(a) for a in (Q, p, G, h, A, b)]
    zhat = dqp.QPFunction(check_Q_spd=True, verbose=-1)(*ins)
    cm = o["best_resid"] < 1e-8
    np.testing.assert_allclose(zhat.detach().cpu().numpy()[cm], o["zhat"][cm], **ZT)
    ct = np.random.default_rng(1).standard_normal((B, 30))
    zhat.backward(dev(ct, grad=False))
    og = oracle.qp_backward(Q, G, A, o["zhat"], o["lam"], o["nu"], o["slack"], ct)
    # The gradient of a QP without strict complementarity (some lam_i ~ slack_i ~ 0) is not
    # well defined: d = clamp(lam)/clamp(slack) (qp.py:149) swings by O(1) on 1e-10 changes of
    # the forward iterate.  Compare where min_i max(lam_i, slack_i) is clearly positive.
    cm &= np.maximum(o["lam"], o["slack"]).min(1) > 1e-5
    assert cm.mean() > 0.95
    for k, t in zip("QpGhAb", ins):
        np.testing.assert_allclose(t.grad.cpu().numpy()[cm], og["d" + k][cm], err_msg="d" + k, **GT)
    # and the backward kernel alone, fed the oracle's forward point, on ALL problems
    from diff_qp_mpc_amd import qp as qpmod
    dv = lambda a: dev(a, grad=False)
    _, _, _, _, _, _, saved = qpmod._forward_impl(*[dv(a) for a in (Q, p, G, h, A, b)], 1e-12, 20, 3)
    gr = qpmod._backward_impl(saved, dv(o["zhat"]), dv(o["lam"]), dv(o["nu"]), dv(o["slack"]), dv(ct),
                              (True,) * 6, qpmod.FORCE_FLAGS)
    for k, t in zip("QpGhAb", gr):
        np.testing.assert_allclose(t.cpu().numpy(), og["d" + k], err_msg="bwd-only d" + k, **GT)


def test_shared_parameters_at_dpp_row_size(dqp):
    """Q, G, A without a batch dim (stride 0 in the C ABI) + .mean(0) gradients (qp.py:160-178)."""
    B = 9
    Q, p, G, h, A, b = family_R(11, B, 30, 30, 15)
    Q0, G0, A0 = Q[0], G[0], A[0]
    z0 = np.random.default_rng(2).standard_normal((B, 30))
    h = z0 @ G0.T + np.random.default_rng(3).random((B, 30))
    b = z0 @ A0.T
    Qe, Ge, Ae = [oracle.expand(a, B, 3) for a in (Q0, G0, A0)]
    o = oracle.qp_forward(Qe, p, Ge, h, Ae, b)
    ins = [dev(a) for a in (Q0, p, G0, h, A0, b)]
    zhat = dqp.QPFunction(check_Q_spd=False, verbose=-1)(*ins)
    np.testing.assert_allclose(zhat.detach().cpu().numpy(), o["zhat"], **ZT)
    zhat.backward(torch.ones_like(zhat))
    og = oracle.qp_backward(Qe, Ge, Ae, o["zhat"], o["lam"], o["nu"], o["slack"], np.ones((B, 30)))
    for k, t in zip("QpGhAb", ins):
        want = og["d" + k].mean(0) if k in "QGA" else og["d" + k]
        np.testing.assert_allclose(t.grad.cpu().numpy(), want, err_msg="d" + k, **GT)


def test_workspace_is_optional(dqp):
    """dqp_qp_forward without a workspace (NULL) runs the equality-row kernels and gives the same
    answer as the null-space path that uses it; dqp_workspace_bytes is what include/dqp.h says."""
    import ctypes
    from diff_qp_mpc_amd import _lib
    lib = _lib.load()
    g = load("R_metric_b8")
    Q, p, G, h, A, b = [dev(g["in_" + k], grad=False) for k in "QpGhAb"]
    B, nz, nineq, neq = 8, 30, 30, 15
    dims = _lib.dqp_dims(B, nz, nineq, neq, nz * nz, nz, nineq * nz, nineq, neq * nz, neq)
    # tails, packed Lq, [Gz | W], U, tau + 1/diag(U), 1/diag(Lq), then xy / py / w1 and delta for the
    # batch rule's finish pass
    per_qp = (15 * 15 + 15 * 14 // 2) + 30 * 31 // 2 + 30 * 30 + 15 * 15 + 2 * 15 + 30 + 3 * 15 + 1
    assert lib.dqp_workspace_bytes(ctypes.byref(dims)) == B * per_qp * 8
    odd = _lib.dqp_dims(B, 7, 5, 2, 49, 7, 35, 5, 14, 2)
    assert lib.dqp_workspace_bytes(ctypes.byref(odd)) == 0         # generic kernels: no scratch
    opts = _lib.dqp_opts(1e-12, 1e-10, 20, 3, 0, 0)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    outs = []
    for use_ws in (False, True):
        kw = dict(dtype=torch.float64, device="cuda")
        zhat, lam, nu, slack = (torch.empty(B, n, **kw) for n in (nz, nineq, neq, nineq))
        info = torch.empty(B, 2, dtype=torch.int32, device="cuda")
        ws = torch.empty(B * per_qp, **kw)
        rc = lib.dqp_qp_forward(ctypes.byref(dims), ctypes.byref(opts), P(Q), P(p), P(G), P(h), P(A),
                                P(b), P(zhat), P(lam), P(nu), P(slack), P(info), None,
                                P(ws) if use_ws else None, None, None)
        assert rc == 0
        torch.cuda.synchronize()
        outs.append([t.cpu().numpy() for t in (zhat, lam, nu, slack)])
        np.testing.assert_allclose(outs[-1][0], g["zhat"], **ZT)
        np.testing.assert_allclose(outs[-1][2], g["nu"], **DT)
    for a0, a1 in zip(*outs):
        np.testing.assert_allclose(a0, a1, rtol=1e-6, atol=1e-8)


def test_not_spd_raises(dqp):
    Q, p, G, h, A, b = family_R(3, 4, 6, 4, 2)
    Q[2] = -Q[2]
    ins = [dev(a, grad=False) for a in (Q, p, G, h, A, b)]
    from diff_qp_mpc_amd import qp as qpmod
    with pytest.raises(RuntimeError, match="Q is not SPD"):
        dqp.QPFunction(check_Q_spd=True, verbose=-1)(*ins)
        qpmod.flush_checks()              # the check is lazy (qp.CHECKS): no synchronisation inside forward


def test_cpu_tensors_fail_loudly(dqp):
    Q, p, G, h, A, b = [torch.tensor(a) for a in family_R(3, 2, 4, 3, 1)]
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dqp.QPFunction(check_Q_spd=False)(Q, p, G, h, A, b)


def test_empty_batch_and_bad_dims(dqp):
    import ctypes
    from diff_qp_mpc_amd import _lib
    lib = _lib.load()
    d = _lib.dqp_dims(4, 513, 3, 0, 0, 0, 0, 0, 0, 0)          # DQP_MAX_DIM_LARGE = 512
    z = ctypes.c_void_p(0)
    assert lib.dqp_qp_forward(ctypes.byref(d), None, *([z] * 15)) == -2      # too large
    d = _lib.dqp_dims(4, 5, 0, 0, 0, 0, 0, 0, 0, 0)
    assert lib.dqp_qp_forward(ctypes.byref(d), None, *([z] * 15)) == -1      # nineq == 0
    d = _lib.dqp_dims(0, 5, 3, 0, 0, 0, 0, 0, 0, 0)
    assert lib.dqp_qp_forward(ctypes.byref(d), None, *([z] * 15)) == 0       # empty batch


@pytest.mark.parametrize("kind", ["M", "R"])
def test_global_batch_rule_over_shards(dqp, kind):
    """The batch-coupled stop over a batch that is split across devices (include/dqp.h:
    dqp_term_local_masks / dqp_qp_forward_finish; sharding.global_batch_rule): two shards solved
    separately with their iteration masks OR-ed in between give bit-identical results to the single solve
    of the whole batch; with shard-local rules they need not (family M: the rule fires early and at
    different iterations per shard)."""
    from diff_qp_mpc_amd import qp as qpmod
    from families import family, family_mpc
    B = 96
    ins = [dev(a, grad=False) for a in (family_mpc(3, B) if kind == "M" else family(3, B, 30, 30, 15, "R"))]
    whole = qpmod._forward_impl(*ins, 1e-12, 20, 3, termination="batch")
    torch.cuda.synchronize()
    cut = 40                                              # ragged shards
    shards = [[t[:cut] for t in ins], [t[cut:] for t in ins]]
    # the exchange: both shards' masks OR-ed.  One process plays both ranks, so the first pass collects the
    # local masks and the second pass hands every shard the combined ones.
    seen = []
    def collect(m):
        seen.append(m.clone())
        return m
    old = qpmod.MASK_EXCHANGE
    try:
        qpmod.MASK_EXCHANGE = collect
        for sh in shards:
            qpmod._forward_impl(*sh, 1e-12, 20, 3, termination="batch")
        combined = seen[0] | seen[1]
        qpmod.MASK_EXCHANGE = lambda m: combined.clone()
        outs = [qpmod._forward_impl(*sh, 1e-12, 20, 3, termination="batch") for sh in shards]
    finally:
        qpmod.MASK_EXCHANGE = old
    torch.cuda.synchronize()
    for k in range(4):                                     # zhat, lam, nu, slack
        got = torch.cat([o[k] for o in outs], 0)
        assert torch.equal(got, whole[k]), "output %d differs from the single-batch solve" % k
    assert torch.equal(torch.cat([o[4] for o in outs], 0)[:, 1], whole[4][:, 1])        # iterations


def test_strict_get_step_flag_freezes_on_exact_zero(dqp):
    """DQP_FLAG_STRICT_GET_STEP (batch.py:211-214 literally: a step component that is exactly 0.0 freezes the problem
    at the best iterate it had).  Which problems meet an exact zero depends on the arithmetic order, so the flag is
    pinned through what it must and must not do on the MPC-structured family (box rows whose multiplier step cancels
    to 0.0 do occur in the kernels' arithmetic too), in the per-problem mode where problems do not influence each
    other: a problem that never meets one is untouched bit for bit; a frozen problem ran fewer iterations and its best
    residual is no better than the default's; the flag fires on a minority of the problems; and under the batch rule
    the forward outputs stay within the test tolerances of the literal oracle wherever the literal and the guarded
    oracle agree (the gradients of a frozen problem move with its weakly active d = lam/slack, as the reference's do:
    tools/strict_vs_guard.py)."""
    from diff_qp_mpc_amd import qp as qpmod, _lib
    B = 2048
    ins_np = family_mpc(0, B)
    dv = [dev(a, grad=False) for a in ins_np]
    base = qpmod.FORCE_FLAGS

    def run(flag, termination):
        qpmod.FORCE_FLAGS = base | flag
        try:
            return qpmod._forward_impl(*dv, 1e-12, 20, 3, termination=termination)
        finally:
            qpmod.FORCE_FLAGS = base
    z0, l0, n0, s0, i0, r0, _ = run(0, "per_problem")
    z1, l1, n1, s1, i1, r1, _ = run(_lib.DQP_FLAG_STRICT_GET_STEP, "per_problem")
    it0, it1 = i0[:, 1].cpu().numpy(), i1[:, 1].cpu().numpy()
    same = ((z0 == z1).all(1) & (l0 == l1).all(1) & (s0 == s1).all(1)).cpu().numpy()
    assert (it1 <= it0).all()
    frozen = it1 < it0
    assert same[~frozen].all()                                   # untouched unless frozen
    assert (r1.cpu().numpy()[frozen] >= r0.cpu().numpy()[frozen]).all()
    assert 1 <= frozen.sum() < B // 2, frozen.sum()
    z2, l2, n2, s2, _, _, _ = run(_lib.DQP_FLAG_STRICT_GET_STEP, "batch")
    o, differ = reference_outputs(ins_np)
    got = {"zhat": z2.cpu().numpy(), "lam": l2.cpu().numpy(), "nu": n2.cpu().numpy(), "slack": s2.cpu().numpy()}
    assert not over_tolerance(got, o)[~differ].any()
