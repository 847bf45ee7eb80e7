"""Pins the CPU oracle (oracle/dqp_oracle.c) to the reference's own outputs.

The fixtures in tests/golden/ were produced by importing the reference
(tests/golden/make_golden.py); here the C restatement must reproduce them.
Tolerances (fp64): zhat/lam/nu/slack rtol 1e-7 atol 1e-9; gradients rtol 1e-5 atol 1e-7 --
both far inside the level at which the reference's two own solvers agree with each other
(3e-13 on zhat, 5e-6 on gradients).
"""
import glob
import os

import numpy as np
import pytest

from oracle import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "[RM]_*.npz")))


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def expanded_inputs(g):
    B = g["zhat"].shape[0]
    out = {}
    for k, nd in (("Q", 3), ("p", 2), ("G", 3), ("h", 2), ("A", 3), ("b", 2)):
        out[k] = oracle.expand(g["in_" + k], B, nd)
    return B, out


def test_fixture_inventory():
    assert len(CASES) >= 10, CASES


@pytest.mark.parametrize("name", CASES)
def test_qp_forward_matches_reference(name):
    g = load(name)
    B, d = expanded_inputs(g)
    o = oracle.qp_forward(d["Q"], d["p"], d["G"], d["h"], d["A"], d["b"])
    np.testing.assert_allclose(o["zhat"], g["zhat"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(o["lam"], g["lam"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(o["slack"], g["slack"], rtol=1e-6, atol=1e-9)
    if g["nu"].shape[1]:
        np.testing.assert_allclose(o["nu"], g["nu"], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("tag", ["ones", "rand"])
def test_qp_backward_matches_reference(name, tag):
    g = load(name)
    B, d = expanded_inputs(g)
    neq = g["nu"].shape[1]
    o = oracle.qp_backward(d["Q"], d["G"], d["A"], g["zhat"], g["lam"], g["nu"], g["slack"],
                           g["ct_" + tag])
    for k, nd in (("Q", 3), ("p", 2), ("G", 3), ("h", 2), ("A", 3), ("b", 2)):
        if neq == 0 and k in "Ab":
            continue
        ref = g["d%s_%s" % (k, tag)]
        got = o["d" + k]
        if g["in_" + k].ndim != nd:          # shared parameter -> .mean(0)  (qp.py:160-178)
            got = got.mean(0)
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-7, err_msg="d" + k)


DENSE = [c for c in CASES if "dense_zhat" in np.load(os.path.join(GOLDEN, c + ".npz")).files]


@pytest.mark.parametrize("name", DENSE)
def test_dense_forward_backward_matches_reference(name):
    g = load(name)
    B, d = expanded_inputs(g)
    o = oracle.dense_forward(d["Q"], d["p"], d["G"], d["h"], d["A"], d["b"])
    np.testing.assert_allclose(o["zhat"], g["dense_zhat"], rtol=1e-7, atol=1e-9)
    for tag in ("ones", "rand"):
        gr = oracle.dense_backward(o["K"], o["zhat"], o["lam"], o["nu"], g["ct_" + tag])
        for k in "QpGhAb":
            np.testing.assert_allclose(gr["d" + k], g["dense_d%s_%s" % (k, tag)],
                                       rtol=1e-5, atol=1e-7, err_msg="dense d" + k)


def test_oracle_thread_count_invariant():
    g = load("R_small_b5")
    B, d = expanded_inputs(g)
    a = oracle.qp_forward(d["Q"], d["p"], d["G"], d["h"], d["A"], d["b"], nthreads=1)
    b = oracle.qp_forward(d["Q"], d["p"], d["G"], d["h"], d["A"], d["b"], nthreads=4)
    assert np.array_equal(a["zhat"], b["zhat"]) and a["iters"] == b["iters"]


# ------------------------------------------------------------------ AL / NewtonAL row
AL_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "AL_*.npz")))


@pytest.mark.parametrize("name", AL_CASES)
def test_al_newton_step_oracle_matches_reference(name):
    from oracle import al_oracle
    g = load(name)
    xu, x0 = g["ns_xu"], g["in_x0"]
    res, resc, J, Jc = al_oracle.constraint_jacobian(xu, x0, g["in_u_lower"], g["in_u_upper"])
    np.testing.assert_allclose(res, g["ns_res"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(resc, g["ns_res_clamp"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(J, g["ns_J"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(Jc, g["ns_Jc"], rtol=1e-12, atol=1e-13)
    B = xu.shape[0]
    lam = np.zeros((B, res.shape[1]))
    grad = al_oracle.merit_grad(xu, g["ns_Qd"], g["in_c"], lam, g["ns_rho"], resc, J, Jc)
    np.testing.assert_allclose(grad, g["ns_grad"], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(al_oracle.hessian(Jc, g["ns_Qd"].reshape(B, -1), g["ns_rho"]), g["ns_H"],
                               rtol=1e-12, atol=1e-12)
    upd, L, info = al_oracle.newton_update(Jc, g["ns_Qd"].reshape(B, -1), g["ns_rho"], g["ns_grad"])
    assert not info.any()
    np.testing.assert_allclose(L, g["ns_L"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(upd, g["ns_update"], rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("tag,guard", [("strict", False), ("guard", True)])
def test_step_rule_variants_match_reference(tag, guard):
    """Mz_guard_b8.npz (make_golden_guard.py): the reference as is, and the reference with
    pdipm_b.get_step replaced by its own batch_LU.get_step (a[dv == 0] = 1).  Samples 1 and 2
    are ones where the unguarded rule divides by an exactly-zero step and freezes the iterate:
    the two reference runs differ by 0.096 in dQ there, and each oracle variant must reproduce
    its own run."""
    g = load("Mz_guard_b8")
    ins = [g["in_" + k] for k in "QpGhAb"]
    o = oracle.qp_forward(*ins, guard=guard)
    np.testing.assert_allclose(o["zhat"], g[tag + "_zhat"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(o["lam"], g[tag + "_lam"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(o["slack"], g[tag + "_slack"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(o["nu"], g[tag + "_nu"], rtol=1e-6, atol=1e-9)
    gr = oracle.qp_backward(ins[0], ins[2], ins[4], o["zhat"], o["lam"], o["nu"], o["slack"], g["ct"])
    for k in "QpGhAb":
        np.testing.assert_allclose(gr["d" + k], g[tag + "_d" + k], rtol=1e-5, atol=1e-7, err_msg="d" + k)
    assert np.abs(g["strict_dQ"] - g["guard_dQ"]).max() > 1e-2      # the fixture does exercise it


def test_broke_down_detector():
    """Late in the solve nearly every family-M sample hits the unguarded division (most of them
    after they converged, which is harmless); with the guard none does, and the guard changes the
    returned iterate only on samples that broke down."""
    from families import broke_down, family_mpc
    ins = family_mpc(0, 256)
    o = oracle.qp_forward(*ins)
    bd = broke_down(o["resid_hist"], o["iters"])
    assert bd.sum() > 0
    o2 = oracle.qp_forward(*ins, guard=True)
    assert not broke_down(o2["resid_hist"], o2["iters"]).any()
    changed = np.abs(o["slack"] - o2["slack"]).max(1) > 1e-9
    assert not (changed & ~bd).any()


# ---------------------------------------------------------------------------------------------------------
# oracle/al_solve_oracle.py (numpy restatement of one AL_mpc.MPC call) against the reference's AL_mpc.MPC outputs:
# x, u are float32 in the reference (AL_mpc.py:319-320) -> rtol 1e-4 / atol 1e-5; multipliers rtol 1e-5 / atol 1e-5
# (cartpole states reach +-pi and controls +-100); rho exact; gradients rtol 1e-4 / atol 1e-6.
def _al_step(robot, dt):
    from oracle import dyn_ref
    if robot in dyn_ref.ROBOTS:
        if not dyn_ref.available(robot):
            pytest.skip("oracle/_ref not built (needs /root/reference; see oracle/Makefile)")

        def step(x, u):
            fx, fu = dyn_ref.jac_x(robot, x, u, dt)
            return dyn_ref.step_x(robot, x, u, dt), fx, fu
        return step
    from oracle import dyn_host                     # host build of the model templates (DYN_*.npz-pinned)
    step = dyn_host.stepper(robot, dt)
    if step is None:
        pytest.skip("hipcc not available")
    return step


@pytest.mark.parametrize("name,robot", [("CFG3_cartpole1l_T20_b4", "cartpole1l"), ("CFG5_cartpole2l_T5_b4", "cartpole2l"),
                                        ("CFG4_rexquadrotor_T6_b4", "rexquadrotor")])
def test_al_solve_oracle_matches_reference(name, robot):
    from oracle import al_solve_oracle as aso
    g = load(name)
    step = _al_step(robot, float(g["dt"]))
    B, T, nt = g["in_Qd"].shape
    n = g["in_x0"].shape[1]
    m = nt - n
    lam0, rho0 = np.zeros((B, T * n + 2 * T * m)), np.ones((B, 1))
    o1 = aso.al_solve(g["in_x_init"], g["in_u_init"], g["in_x0"], g["in_Qd"], g["in_c"], g["in_u_lower"], g["in_u_upper"],
                      step, lam0, rho0)
    np.testing.assert_allclose(o1["x"], g["x1"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(o1["u"], g["u1"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(o1["lam"], g["lam1"], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(o1["rho"], g["rho1"])
    # backward of x.sum() + 2 u.sum() (the generator's loss)
    gxu = np.concatenate((np.ones((B, T, n)), 2.0 * np.ones((B, T, m))), 2)
    dQ, dq = aso.backward(o1["L"], o1["xu"], gxu)
    np.testing.assert_allclose(dQ, g["dC1"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(dq, g["dc1"], rtol=1e-4, atol=1e-6)
    # the warm-started second call starts from the float32 solution of the first (AL_mpc.py:250-251)
    o2 = aso.al_solve(g["x1"].astype(np.float64), g["u1"].astype(np.float64), g["in_x0"], g["in_Qd"], g["in_c"],
                      g["in_u_lower"], g["in_u_upper"], step, o1["lam"], o1["rho"], history=o1["history"])
    np.testing.assert_allclose(o2["x"], g["x2"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(o2["u"], g["u2"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(o2["lam"], g["lam2"], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(o2["rho"], g["rho2"])
