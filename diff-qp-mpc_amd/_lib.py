"""ctypes binding of the C ABI in include/dqp.h (csrc/libdqp_hip.so).

The library is loaded lazily on first use and there is deliberately no fallback: a missing
or unloadable library raises RuntimeError naming the build command.
"""
import ctypes
import os

from . import _build

_dp = ctypes.c_void_p


class dqp_dims(ctypes.Structure):
    _fields_ = [("nbatch", ctypes.c_int32), ("nz", ctypes.c_int32), ("nineq", ctypes.c_int32),
                ("neq", ctypes.c_int32),
                ("stride_Q", ctypes.c_int64), ("stride_p", ctypes.c_int64),
                ("stride_G", ctypes.c_int64), ("stride_h", ctypes.c_int64),
                ("stride_A", ctypes.c_int64), ("stride_b", ctypes.c_int64)]


class dqp_opts(ctypes.Structure):
    _fields_ = [("eps", ctypes.c_double), ("stall_tol", ctypes.c_double),
                ("max_iter", ctypes.c_int32),
                ("not_improved_lim", ctypes.c_int32), ("flags", ctypes.c_uint32),
                ("reserved", ctypes.c_int32),
                ("dyn_id", ctypes.c_int32), ("dyn_T", ctypes.c_int32), ("dyn_dt", ctypes.c_double),
                ("dyn_x0", ctypes.c_void_p)]


DQP_OK = 0
DQP_FLAG_DENSE_BACKWARD = 1
DQP_FLAG_GENERIC_ONLY = 2
DQP_FLAG_NO_NULLSPACE = 4
DQP_FLAG_RIC_GLOBAL_WS = 64
DQP_FLAG_BACKWARD_CTX = 8
DQP_FLAG_BATCH_TERMINATION = 16
DQP_FLAG_HISTORY_ONLY = 32
DQP_FLAG_STRICT_GET_STEP = 128
DQP_FLAG_STAGEWISE = 256
DQP_STATUS_Q_NOT_PD = 1
DQP_STATUS_A_RANK_DEF = 2
DQP_MAX_DIM = 64
DQP_MAX_DIM_LARGE = 512

# every symbol include/dqp.h declares
SYMBOLS = ("dqp_version", "dqp_error_string", "dqp_workspace_bytes", "dqp_termination_bytes",
           "dqp_qp_forward", "dqp_qp_backward", "dqp_term_local_masks", "dqp_qp_forward_finish", "dqp_mpc_assemble", "dqp_mpc_assemble_backward",
           "dqp_mpc_qp_supported", "dqp_mpc_qp_workspace_bytes", "dqp_mpc_qp_termination_bytes", "dqp_mpc_qp_forward", "dqp_mpc_qp_backward", "dqp_mpc_line_search", "dqp_mpc_rollout_backward",
           "dqp_al_newton_step", "dqp_al_chol_solve", "dqp_al_assemble", "dqp_al_merit",
           "dqp_al_newton_solve_bytes", "dqp_al_newton_solve",
           "dqp_al_outer_update", "dqp_al_banded_factor_bytes", "dqp_al_banded_newton_step", "dqp_al_banded_solve",
           "dqp_al_banded_newton_step_jac", "dqp_al_lane_group", "dqp_al_mpc_solve_bytes", "dqp_al_mpc_solve",
           "dqp_mpc_qp_stepped_workspace_bytes", "dqp_mpc_qp_stepped_termination_bytes", "dqp_mpc_qp_forward_stepped", "dqp_trace_begin", "dqp_trace_end",
           "dqp_dyn_sizes", "dqp_dyn_step", "dqp_dyn_jacobian", "dqp_dyn_forward_dynamics",
           "dqp_dyn_forward_derivatives")
DQP_DYN = {"pendulum1l": 1, "cartpole1l": 2, "cartpole2l": 3, "pendulum_euler": 4, "pendulum_dx": 5,
           "rexquadrotor": 6}


class dqp_al_mpc_dims(ctypes.Structure):
    _fields_ = [("nbatch", ctypes.c_int32), ("n_state", ctypes.c_int32), ("n_ctrl", ctypes.c_int32),
                ("T", ctypes.c_int32)]


class dqp_al_dims(ctypes.Structure):
    _fields_ = [("nbatch", ctypes.c_int32), ("nz", ctypes.c_int32), ("ncon", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class dqp_mpc_dims(ctypes.Structure):
    _fields_ = [("nbatch", ctypes.c_int32), ("n_state", ctypes.c_int32), ("n_ctrl", ctypes.c_int32),
                ("T", ctypes.c_int32), ("has_bounds", ctypes.c_int32), ("dyn_id", ctypes.c_int32)]



class dqp_trace_record(ctypes.Structure):
    _fields_ = [("kernel", ctypes.c_char * 248), ("ms", ctypes.c_float), ("reserved", ctypes.c_int32)]


_lib = None


def path():
    # DQP_HIP_LIBRARY: A/B runs of an alternative build of the same library (tools/, profiling)
    return os.environ.get("DQP_HIP_LIBRARY") or _build.SO


def load():
    """Returns the ctypes CDLL; raises if the HIP library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    so = path()
    if not os.path.exists(so):
        raise RuntimeError(
            "diff_qp_mpc_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % so)
    try:
        lib = ctypes.CDLL(so)
    except OSError as e:  # pragma: no cover
        raise RuntimeError("diff_qp_mpc_amd: cannot load %s: %s" % (so, e))
    lib.dqp_version.restype = ctypes.c_int
    lib.dqp_error_string.restype = ctypes.c_char_p
    lib.dqp_error_string.argtypes = [ctypes.c_int]
    lib.dqp_workspace_bytes.restype = ctypes.c_size_t
    lib.dqp_workspace_bytes.argtypes = [ctypes.POINTER(dqp_dims)]
    lib.dqp_qp_forward.restype = ctypes.c_int
    lib.dqp_termination_bytes.restype = ctypes.c_size_t
    lib.dqp_termination_bytes.argtypes = [ctypes.POINTER(dqp_dims), ctypes.POINTER(dqp_opts)]
    lib.dqp_qp_forward.argtypes = [ctypes.POINTER(dqp_dims), ctypes.POINTER(dqp_opts)] + [_dp] * 15
    lib.dqp_term_local_masks.restype = ctypes.c_int
    lib.dqp_term_local_masks.argtypes = [ctypes.POINTER(dqp_dims), ctypes.POINTER(dqp_opts)] + [_dp] * 3
    lib.dqp_qp_forward_finish.restype = ctypes.c_int
    lib.dqp_qp_forward_finish.argtypes = [ctypes.POINTER(dqp_dims), ctypes.POINTER(dqp_opts)] + [_dp] * 16
    lib.dqp_qp_backward.restype = ctypes.c_int
    lib.dqp_qp_backward.argtypes = [ctypes.POINTER(dqp_dims), ctypes.POINTER(dqp_opts)] + [_dp] * 17
    lib.dqp_mpc_assemble.restype = ctypes.c_int
    lib.dqp_mpc_assemble.argtypes = [ctypes.POINTER(dqp_mpc_dims)] + [_dp] * 14
    lib.dqp_mpc_assemble_backward.restype = ctypes.c_int
    lib.dqp_mpc_assemble_backward.argtypes = [ctypes.POINTER(dqp_mpc_dims)] + [_dp] * 10
    lib.dqp_mpc_qp_supported.restype = ctypes.c_int
    lib.dqp_mpc_qp_supported.argtypes = [ctypes.POINTER(dqp_mpc_dims)]
    lib.dqp_mpc_qp_workspace_bytes.restype = ctypes.c_size_t
    lib.dqp_mpc_qp_workspace_bytes.argtypes = [ctypes.POINTER(dqp_mpc_dims)]
    lib.dqp_mpc_qp_termination_bytes.restype = ctypes.c_size_t
    lib.dqp_mpc_qp_termination_bytes.argtypes = [ctypes.POINTER(dqp_mpc_dims), ctypes.POINTER(dqp_opts)]
    lib.dqp_mpc_qp_forward.restype = ctypes.c_int
    lib.dqp_mpc_qp_forward.argtypes = [ctypes.POINTER(dqp_mpc_dims), ctypes.POINTER(dqp_opts)] + [_dp] * 16
    lib.dqp_mpc_qp_backward.restype = ctypes.c_int
    lib.dqp_mpc_qp_backward.argtypes = [ctypes.POINTER(dqp_mpc_dims), ctypes.POINTER(dqp_opts)] + [_dp] * 15
    lib.dqp_mpc_line_search.restype = ctypes.c_int
    lib.dqp_mpc_line_search.argtypes = ([ctypes.POINTER(dqp_mpc_dims), ctypes.c_int, ctypes.c_double] + [_dp] * 8 +
                                        [ctypes.c_double, ctypes.c_int32] + [_dp] * 5)
    lib.dqp_mpc_rollout_backward.restype = ctypes.c_int
    lib.dqp_mpc_rollout_backward.argtypes = [ctypes.POINTER(dqp_mpc_dims), ctypes.c_int, ctypes.c_double] + [_dp] * 9
    lib.dqp_al_newton_step.restype = ctypes.c_int
    lib.dqp_al_newton_step.argtypes = [ctypes.POINTER(dqp_al_dims)] + [_dp] * 8
    lib.dqp_al_chol_solve.restype = ctypes.c_int
    lib.dqp_al_chol_solve.argtypes = [ctypes.POINTER(dqp_al_dims)] + [_dp] * 4
    lib.dqp_al_assemble.restype = ctypes.c_int
    lib.dqp_al_assemble.argtypes = [ctypes.POINTER(dqp_al_mpc_dims)] + [_dp] * 8
    lib.dqp_al_merit.restype = ctypes.c_int
    lib.dqp_al_merit.argtypes = [ctypes.POINTER(dqp_al_mpc_dims), ctypes.c_int32] + [_dp] * 11
    lib.dqp_al_newton_solve_bytes.restype = ctypes.c_size_t
    lib.dqp_al_newton_solve_bytes.argtypes = [ctypes.POINTER(dqp_al_mpc_dims), ctypes.c_int32]
    lib.dqp_al_newton_solve.restype = ctypes.c_int
    lib.dqp_al_newton_solve.argtypes = ([ctypes.POINTER(dqp_al_mpc_dims), ctypes.c_int, ctypes.c_double, ctypes.c_int32,
                                         ctypes.c_int32] + [_dp] * 13)
    lib.dqp_al_outer_update.restype = ctypes.c_int
    lib.dqp_al_outer_update.argtypes = [ctypes.POINTER(dqp_al_mpc_dims), ctypes.c_int, ctypes.c_double] + [_dp] * 12
    lib.dqp_al_banded_factor_bytes.restype = ctypes.c_size_t
    lib.dqp_al_banded_factor_bytes.argtypes = [ctypes.POINTER(dqp_al_mpc_dims), ctypes.c_int]
    lib.dqp_al_banded_newton_step.restype = ctypes.c_int
    lib.dqp_al_banded_newton_step.argtypes = [ctypes.POINTER(dqp_al_mpc_dims), ctypes.c_int, ctypes.c_double] + [_dp] * 12
    lib.dqp_al_banded_newton_step_jac.restype = ctypes.c_int
    lib.dqp_al_banded_newton_step_jac.argtypes = [ctypes.POINTER(dqp_al_mpc_dims)] + [_dp] * 15
    lib.dqp_al_banded_solve.restype = ctypes.c_int
    lib.dqp_al_banded_solve.argtypes = [ctypes.POINTER(dqp_al_mpc_dims), ctypes.c_int] + [_dp] * 4
    lib.dqp_al_mpc_solve_bytes.restype = ctypes.c_size_t
    lib.dqp_al_mpc_solve_bytes.argtypes = [ctypes.POINTER(dqp_al_mpc_dims)]
    lib.dqp_al_mpc_solve.restype = ctypes.c_int
    lib.dqp_al_mpc_solve.argtypes = ([ctypes.POINTER(dqp_al_mpc_dims), ctypes.c_int, ctypes.c_double, ctypes.c_int32, ctypes.c_int32]
                                     + [_dp] * 12 + [ctypes.c_int32] + [_dp] * 10)
    lib.dqp_al_lane_group.restype = ctypes.c_int
    lib.dqp_al_lane_group.argtypes = [ctypes.c_int]
    lib.dqp_mpc_qp_stepped_workspace_bytes.restype = ctypes.c_size_t
    lib.dqp_mpc_qp_stepped_workspace_bytes.argtypes = [ctypes.POINTER(dqp_mpc_dims)]
    lib.dqp_mpc_qp_stepped_termination_bytes.restype = ctypes.c_size_t
    lib.dqp_mpc_qp_stepped_termination_bytes.argtypes = [ctypes.POINTER(dqp_mpc_dims), ctypes.POINTER(dqp_opts)]
    lib.dqp_mpc_qp_forward_stepped.restype = ctypes.c_int
    lib.dqp_mpc_qp_forward_stepped.argtypes = ([ctypes.POINTER(dqp_mpc_dims), ctypes.POINTER(dqp_opts)] + [_dp] * 8 +
                                               [ctypes.c_int32, ctypes.c_int32] + [_dp] * 9)
    lib.dqp_trace_begin.restype = ctypes.c_int
    lib.dqp_trace_begin.argtypes = [ctypes.c_int32]
    lib.dqp_trace_end.restype = ctypes.c_int
    lib.dqp_trace_end.argtypes = [ctypes.POINTER(dqp_trace_record), ctypes.c_int32, ctypes.POINTER(ctypes.c_int32)]
    i32p = ctypes.POINTER(ctypes.c_int32)
    lib.dqp_dyn_sizes.restype = ctypes.c_int
    lib.dqp_dyn_sizes.argtypes = [ctypes.c_int, i32p, i32p]
    lib.dqp_dyn_step.restype = ctypes.c_int
    lib.dqp_dyn_step.argtypes = [ctypes.c_int, ctypes.c_int32, _dp, _dp, ctypes.c_double, _dp, _dp]
    lib.dqp_dyn_jacobian.restype = ctypes.c_int
    lib.dqp_dyn_jacobian.argtypes = [ctypes.c_int, ctypes.c_int32, _dp, _dp, ctypes.c_double] + [_dp] * 4
    lib.dqp_dyn_forward_dynamics.restype = ctypes.c_int
    lib.dqp_dyn_forward_dynamics.argtypes = [ctypes.c_int, ctypes.c_int32] + [_dp] * 7
    lib.dqp_dyn_forward_derivatives.restype = ctypes.c_int
    lib.dqp_dyn_forward_derivatives.argtypes = [ctypes.c_int, ctypes.c_int32] + [_dp] * 11
    _lib = lib
    return lib


def check(rc, what):
    if rc != DQP_OK:
        msg = load().dqp_error_string(rc).decode()
        raise RuntimeError("diff_qp_mpc_amd: %s failed: %s (code %d)" % (what, msg, rc))


class trace:
    """with _lib.trace(max_launches) as t: ...  -> t.records = [(kernel name, ms)], t.by_kernel() = {name: (count, mean ms)}:
    HIP-event timing of every kernel launch the library makes inside the block (dqp_trace_begin / dqp_trace_end)."""

    def __init__(self, max_launches=4096):
        self.cap = int(max_launches)
        self.records = []

    def __enter__(self):
        check(load().dqp_trace_begin(self.cap), "dqp_trace_begin")
        return self

    def __exit__(self, *exc):
        buf = (dqp_trace_record * self.cap)()
        n = ctypes.c_int32(0)
        rc = load().dqp_trace_end(buf, self.cap, ctypes.byref(n))
        self.records = [(buf[i].kernel.decode(errors="replace"), float(buf[i].ms)) for i in range(min(n.value, self.cap))]
        if exc[0] is None:
            check(rc, "dqp_trace_end")
        return False

    def by_kernel(self):
        acc = {}
        for k, ms in self.records:
            c, t = acc.get(k, (0, 0.0))
            acc[k] = (c + 1, t + ms)
        return {k: (c, t / c) for k, (c, t) in acc.items()}
