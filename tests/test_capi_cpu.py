"""CPU-side checks of the drop-in boundary: the HIP library was built, loads, and exports
every symbol include/dqp.h declares; host-side argument validation works without a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from diff_qp_mpc_amd import _lib
    return _lib.load()


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "dqp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dqp_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    from diff_qp_mpc_amd import _lib
    syms = declared_symbols()
    assert set(syms) == set(_lib.SYMBOLS), (syms, _lib.SYMBOLS)
    for s in syms:
        assert hasattr(lib, s), s


def test_version_and_error_strings(lib):
    assert lib.dqp_version() >= 100
    assert lib.dqp_error_string(0) == b"ok"
    assert b"large" in lib.dqp_error_string(-2)


def test_argument_validation_without_gpu(lib):
    from diff_qp_mpc_amd import _lib
    z = ctypes.c_void_p(0)
    d = _lib.dqp_dims(4, 513, 3, 0, 0, 0, 0, 0, 0, 0)          # DQP_MAX_DIM_LARGE = 512
    assert lib.dqp_qp_forward(ctypes.byref(d), None, *([z] * 15)) == -2
    d = _lib.dqp_dims(4, 5, 3, 2, 0, 0, 0, 0, 0, 0)
    assert lib.dqp_qp_forward(ctypes.byref(d), None, *([z] * 15)) == -1   # null pointers
    assert lib.dqp_qp_backward(ctypes.byref(d), None, *([z] * 17)) == -1
    d = _lib.dqp_dims(0, 5, 3, 2, 0, 0, 0, 0, 0, 0)
    assert lib.dqp_qp_forward(ctypes.byref(d), None, *([z] * 15)) == 0    # empty batch: no launch


def test_workspace_bytes_is_a_host_function(lib):
    """include/dqp.h: optional workspace per QP = reflector tails + the factorisation context
    (packed Lq, [Gz | W], U, tau, 1/diag U, 1/diag Lq) for the sizes that have a null-space
    kernel, 0 otherwise (generic kernels, no equalities)."""
    from diff_qp_mpc_amd import _lib, _build
    for nz, nineq, neq in _build.R16N_SIZES:
        d = _lib.dqp_dims(7, nz, nineq, neq, 0, 0, 0, 0, 0, 0)
        per_qp = (neq * (nz - neq) + neq * (neq - 1) // 2) + nz * (nz + 1) // 2 + nineq * nz + neq * neq + 5 * neq + nz + 1
        assert lib.dqp_workspace_bytes(ctypes.byref(d)) == 7 * per_qp * 8
    for nz, nineq, neq in [(12, 8, 0), (7, 5, 2), (64, 64, 32)]:
        d = _lib.dqp_dims(7, nz, nineq, neq, 0, 0, 0, 0, 0, 0)
        assert lib.dqp_workspace_bytes(ctypes.byref(d)) == 0
    d = _lib.dqp_dims(0, 30, 30, 15, 0, 0, 0, 0, 0, 0)
    assert lib.dqp_workspace_bytes(ctypes.byref(d)) == 0
    assert lib.dqp_workspace_bytes(None) == 0


def test_mpc_argument_validation_without_gpu(lib):
    from diff_qp_mpc_amd import _lib
    z = ctypes.c_void_p(0)
    d = _lib.dqp_mpc_dims(4, 3, 3, 1, 1, 0)                       # T < 2
    assert lib.dqp_mpc_assemble(ctypes.byref(d), *([z] * 15)) == -1
    d = _lib.dqp_mpc_dims(4, 3, 3, 5, 1, 0)                       # null pointers
    assert lib.dqp_mpc_assemble(ctypes.byref(d), *([z] * 15)) == -1
    d = _lib.dqp_mpc_dims(0, 3, 3, 5, 1, 0)                       # empty batch
    assert lib.dqp_mpc_assemble(ctypes.byref(d), *([z] * 15)) == 0
    assert lib.dqp_mpc_assemble_backward(ctypes.byref(d), *([z] * 10)) == 0


def test_operators_refuse_cpu_tensors():
    import torch
    import diff_qp_mpc_amd as dqp
    Q = torch.eye(3, dtype=torch.float64).unsqueeze(0)
    p = torch.zeros(1, 3, dtype=torch.float64)
    G = torch.ones(1, 2, 3, dtype=torch.float64)
    h = torch.ones(1, 2, dtype=torch.float64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dqp.QPFunction(check_Q_spd=False)(Q, p, G, h, torch.Tensor(), torch.Tensor())


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "diff-qp-mpc_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("offline oracle", ""), os.path.join(dp, f)


def test_header_constants_match_the_binding(lib):
    """Every DQP_FLAG_* / DQP_DYN_* value of include/dqp.h is the one the ctypes binding uses, the flags are
    distinct bits, and the library reports the header's version."""
    from diff_qp_mpc_amd import _lib
    src = open(os.path.join(ROOT, "include", "dqp.h")).read()
    flags = {k: int(v) for k, v in re.findall(r"#define\s+(DQP_FLAG_[A-Z_]+)\s+(\d+)u", src)}
    assert len(flags) >= 7
    for k, v in flags.items():
        assert getattr(_lib, k) == v, k
        assert v & (v - 1) == 0, k
    assert len(set(flags.values())) == len(flags)
    dyn = {k: int(v) for k, v in re.findall(r"(DQP_DYN_[A-Z0-9_]+)\s*=\s*(\d+)", src)}
    assert dyn.get("DQP_DYN_REXQUADROTOR") == 6
    version = int(re.search(r"#define\s+DQP_VERSION\s+(\d+)", src).group(1))
    assert lib.dqp_version() == version


def test_stagewise_workspace_bytes_is_a_host_function(lib):
    """dqp_mpc_qp_workspace_bytes for the stage-wise kernels: per problem the iterate, residual, direction
    and best-iterate vectors, the factor rows as P_t (n x n) and [Lxu ; Luu] (nt x m), p_t and Luu^-1 h_u;
    four problems per wavefront, so the batch is rounded up to a multiple of four."""
    from diff_qp_mpc_amd import _lib
    for n, m, T, B in [(12, 4, 30, 8192), (12, 4, 30, 5), (6, 1, 40, 9), (3, 1, 30, 1024)]:        # nz > 64: no dense kernel
        nt = n + m
        it = T * (nt + n + 4 * m)
        per = it + T * (n + 2 * m) + it + it          # X..ZL | RY RZU RZL | DX..DZL | BX..BZL
        per += per & 1
        per += T * n * n
        per += per & 1
        per += T * nt * m + T * n + T * m
        per += per & 1
        d = _lib.dqp_mpc_dims(B, n, m, T, 1, 0)
        assert lib.dqp_mpc_qp_supported(ctypes.byref(d)) == 1
        assert lib.dqp_mpc_qp_workspace_bytes(ctypes.byref(d)) == (B + 3) // 4 * 4 * per * 8, (n, m, T, B)


def test_stagewise_horizon_limit(lib):
    """The stage-wise kernels address a wavefront's workspaces with 32-bit byte offsets: a horizon whose four
    workspaces exceed 2 GB is refused on the host (DQP_ERR_TOO_LARGE = -2), not launched."""
    from diff_qp_mpc_amd import _lib
    d = _lib.dqp_mpc_dims(8, 12, 4, 190000, 1, 0)
    assert lib.dqp_mpc_qp_supported(ctypes.byref(d)) == 0
    assert lib.dqp_mpc_qp_workspace_bytes(ctypes.byref(d)) == 0
    d = _lib.dqp_mpc_dims(8, 12, 4, 2000, 1, 0)
    assert lib.dqp_mpc_qp_supported(ctypes.byref(d)) == 1
