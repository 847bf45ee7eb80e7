"""Host build of the dynamics model templates (diff-qp-mpc_amd/csrc/dqp_dyn_models.h instantiated by
tests/host/dyn_host.cpp) as a CPU `step` for the numpy AL oracle: x_next and both Jacobians of every registered
model, pinned against the reference's outputs by tests/test_dynamics_cpu.py (DYN_*.npz).

TEST INFRASTRUCTURE ONLY (tests/, bench.py's cpu_baseline leg): nothing under diff-qp-mpc_amd/ imports it.
"""
import ctypes
import os
import shutil
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
IDS = {"pendulum1l": 1, "cartpole1l": 2, "cartpole2l": 3, "pendulum_euler": 4, "pendulum_dx": 5, "rexquadrotor": 6}
_lib = None


def build():
    """tests/host/dyn_host.cpp -> tests/host/_build/libdyn_host.so; None without hipcc and without a current
    prebuilt library."""
    global _lib
    if _lib is not None:
        return _lib
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    out = os.path.join(_ROOT, "tests", "host", "_build")
    so = os.path.join(out, "libdyn_host.so")
    src = os.path.join(_ROOT, "tests", "host", "dyn_host.cpp")
    hdr = os.path.join(_ROOT, "diff-qp-mpc_amd", "csrc", "dqp_dyn_models.h")
    stale = not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    if stale:
        if not os.path.exists(hipcc):
            return None
        os.makedirs(out, exist_ok=True)
        subprocess.check_call([hipcc, "-x", "hip", "--cuda-host-only", "-O2", "-std=c++17", "-fPIC", "-shared",
                               "-ffp-contract=off", "-o", so, src])
    lib = ctypes.CDLL(so)
    lib.dyn_host_jac.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 2 + [ctypes.c_double] + \
        [ctypes.c_void_p] * 3
    _lib = lib
    return lib


def stepper(robot, dt):
    """-> step(x (N,n), u (N,m)) -> (x_next, df/dx (N,n,n), df/du (N,n,m)), or None when the library cannot be built."""
    lib = build()
    if lib is None:
        return None
    rid = IDS[robot]

    def step(x, u):
        x, u = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(u, dtype=np.float64)
        N, n, m = x.shape[0], x.shape[1], u.shape[1]
        xn, Jx, Ju = np.empty((N, n)), np.empty((N, n, n)), np.empty((N, n, m))
        if lib.dyn_host_jac(rid, N, x.ctypes.data, u.ctypes.data, dt, xn.ctypes.data, Jx.ctypes.data, Ju.ctypes.data) != 0:
            raise RuntimeError("dyn_host_jac: unknown model %s" % robot)
        return xn, Jx, Ju
    return step
