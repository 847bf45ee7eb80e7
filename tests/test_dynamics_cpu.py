"""CPU-side checks of the device dynamics registry.

  * the model templates the HIP kernels inline (csrc/dqp_dyn_models.h), instantiated for the host
    by tests/host/dyn_host.cpp, against the golden vectors of tests/golden/DYN_*.npz -- outputs of
    the reference's CasADi-generated C (robots) and of its torch modules (pendulums):
    states 1e-12, Jacobians 1e-11 (absolute; values are O(1-10));
  * oracle/_ref (the reference's generated C compiled where it lies) against the same goldens,
    when it has been built (build container; it also travels to the GPU box).
"""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import dyn_ref

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
IDS = {"pendulum1l": 1, "cartpole1l": 2, "cartpole2l": 3, "pendulum_euler": 4, "pendulum_dx": 5, "rexquadrotor": 6}


def build_hostlib():
    from oracle import dyn_host
    return dyn_host.build()


@pytest.fixture(scope="module")
def hostlib():
    lib = build_hostlib()
    if lib is None:
        pytest.skip("hipcc not available")
    return lib


@pytest.mark.parametrize("name", sorted(IDS))
def test_model_templates_match_reference(hostlib, name):
    g = np.load(os.path.join(GOLDEN, "DYN_%s.npz" % name))
    x, u = np.ascontiguousarray(g["x"]), np.ascontiguousarray(g["u"])
    N, n = x.shape
    m = u.shape[1]
    xn, Jx, Ju = np.empty((N, n)), np.empty((N, n, n)), np.empty((N, n, m))
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert hostlib.dyn_host_jac(IDS[name], N, P(x), P(u), float(g["dt"]), P(xn), P(Jx), P(Ju)) == 0
    np.testing.assert_allclose(xn, g["x_next"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(Jx, g["Jx"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(Ju, g["Ju"], rtol=0, atol=1e-11)


@pytest.mark.parametrize("robot", sorted(dyn_ref.ROBOTS))
def test_ref_build_matches_goldens(robot):
    if not dyn_ref.available(robot):
        pytest.skip("oracle/_ref not built (needs the reference checkout: make -C oracle ref)")
    g = np.load(os.path.join(GOLDEN, "DYN_%s.npz" % robot))
    qo, qdo = dyn_ref.dynamics(robot, g["q"], g["qd"], g["tau"], g["h"])
    np.testing.assert_allclose(qo, g["q_out"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(qdo, g["qd_out"], rtol=0, atol=1e-13)
    for i, b in enumerate(dyn_ref.derivatives(robot, g["q"], g["qd"], g["tau"], g["h"])):
        np.testing.assert_allclose(b, g["blk%d" % i], rtol=0, atol=1e-13)
