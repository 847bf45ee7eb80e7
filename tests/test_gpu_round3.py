"""Round-3 additions to the C ABI, on the GPU:
  * dqp_trace_begin / dqp_trace_end: one record per launch, names and plausible times;
  * dqp_mpc_qp_forward_stepped (one PDIPM iteration per call, the equality residual supplied by the caller) against
    dqp_mpc_qp_forward on the same stage-wise kernels (DQP_FLAG_STAGEWISE) with the linear residual computed in
    torch: tau / duals rtol 1e-6 / atol 1e-8 (the two residuals differ by round-off only), both termination modes;
  * DQP_FLAG_STAGEWISE on a shape that also has a null-space kernel: same solution from both families.
"""
import ctypes

import numpy as np
import pytest
import torch

from test_gpu_ric import dev, problem

pytestmark = pytest.mark.gpu


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _mpc_buffers(n, m, T, B):
    kw = dict(dtype=torch.float64, device="cuda")
    return dict(tau=torch.empty(B, T, n + m, **kw), lam=torch.empty(B, 2 * T * m, **kw), nu=torch.empty(B, T * n, **kw),
                slack=torch.empty(B, 2 * T * m, **kw), info=torch.empty(B, 2, dtype=torch.int32, device="cuda"),
                resid=torch.empty(B, **kw))


def _forward(lib, _lib, n, m, T, data, flags, stepped=False):
    C, c, F, f, x0, lo, hi = [dev(a) for a in data]
    B = x0.shape[0]
    dims = _lib.dqp_mpc_dims(B, n, m, T, 1, 0)
    opts = _lib.dqp_opts(1e-12, 1e-10, 20, 3, flags, 0)
    o = _mpc_buffers(n, m, T, B)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    kw = dict(dtype=torch.float64, device="cuda")
    ins = [_p(t) for t in (C, c, F, f, x0, lo, hi)]
    outs = [_p(o[k]) for k in ("tau", "lam", "nu", "slack", "info", "resid")]
    if not stepped:
        ws = torch.empty(int(lib.dqp_mpc_qp_workspace_bytes(ctypes.byref(dims))) // 8 +
                         int(lib.dqp_mpc_qp_stepped_workspace_bytes(ctypes.byref(dims))) // 8, **kw)
        tb = int(lib.dqp_mpc_qp_termination_bytes(ctypes.byref(dims), ctypes.byref(opts)))
        tb = max(tb, int(lib.dqp_mpc_qp_stepped_termination_bytes(ctypes.byref(dims), ctypes.byref(opts))))
        term = torch.empty(max(tb // 8 + 1, 1), **kw)
        assert lib.dqp_mpc_qp_forward(ctypes.byref(dims), ctypes.byref(opts), *ins, *outs, _p(ws), _p(term), st) == 0
        return o
    ws = torch.empty(int(lib.dqp_mpc_qp_stepped_workspace_bytes(ctypes.byref(dims))) // 8, **kw)
    tb = int(lib.dqp_mpc_qp_stepped_termination_bytes(ctypes.byref(dims), ctypes.byref(opts)))
    term = torch.empty(max(tb // 8 + 1, 1), **kw)
    call = lambda ry, a, b: lib.dqp_mpc_qp_forward_stepped(ctypes.byref(dims), ctypes.byref(opts), *ins, _p(ry), a, b, *outs,
                                                           _p(ws), _p(term), st)
    assert call(None, 0, 0) == 0
    for it in range(20):
        tau = o["tau"]
        pred = torch.matmul(F.transpose(0, 1), tau[:, :-1, :, None])[..., 0] + f.transpose(0, 1)
        ry = torch.cat(((pred - tau[:, 1:, :n]).reshape(B, -1), tau[:, 0, :n] - x0), 1).contiguous()
        assert call(ry, it, it + 1) == 0
    return o


@pytest.mark.parametrize("n,m,T,B", [(3, 3, 5, 37), (3, 1, 10, 6), (12, 4, 6, 5)])
@pytest.mark.parametrize("batch_rule", [True, False])
def test_stepped_forward_matches_fused_stagewise(n, m, T, B, batch_rule):
    from diff_qp_mpc_amd import _lib
    lib = _lib.load()
    data = problem(n, m, T, B, seed=7 + n)
    flags = _lib.DQP_FLAG_STAGEWISE | (_lib.DQP_FLAG_BATCH_TERMINATION if batch_rule else 0)
    a = _forward(lib, _lib, n, m, T, data, flags)
    b = _forward(lib, _lib, n, m, T, data, flags, stepped=True)
    torch.cuda.synchronize()
    assert int(a["info"][:, 0].abs().max()) == 0 and int(b["info"][:, 0].abs().max()) == 0
    np.testing.assert_allclose(b["tau"].cpu().numpy(), a["tau"].cpu().numpy(), rtol=1e-6, atol=1e-8)
    for k in ("lam", "nu", "slack"):
        np.testing.assert_allclose(b[k].cpu().numpy(), a[k].cpu().numpy(), rtol=1e-5, atol=1e-7)
    if batch_rule:      # the caller-driven loop replays the same rule: same stop iteration I* (the smallest count; which
        # problems are taken back to it depends on round-off level differences of their late residuals)
        assert int(b["info"][:, 1].min()) == int(a["info"][:, 1].min())


def test_stepped_argument_checks():
    from diff_qp_mpc_amd import _lib
    lib = _lib.load()
    n, m, T, B = 3, 1, 5, 4
    C, c, F, f, x0, lo, hi = [dev(a) for a in problem(n, m, T, B, seed=1)]
    dims = _lib.dqp_mpc_dims(B, n, m, T, 1, 0)
    opts = _lib.dqp_opts(1e-12, 1e-10, 20, 3, 0, 0)
    o = _mpc_buffers(n, m, T, B)
    ws = torch.empty(int(lib.dqp_mpc_qp_stepped_workspace_bytes(ctypes.byref(dims))) // 8, dtype=torch.float64, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    base = [ctypes.byref(dims), ctypes.byref(opts)] + [_p(t) for t in (C, c, F, f, x0, lo, hi)]
    outs = [_p(o[k]) for k in ("tau", "lam", "nu", "slack", "info", "resid")] + [_p(ws), _p(None), st]
    ry = torch.zeros(B, T * n, dtype=torch.float64, device="cuda")
    assert lib.dqp_mpc_qp_forward_stepped(*base, _p(None), 3, 4, *outs) == -1        # an iteration needs its residual
    assert lib.dqp_mpc_qp_forward_stepped(*base, _p(ry), 0, 2, *outs) == -1          # one iteration per call
    assert lib.dqp_mpc_qp_forward_stepped(*base, _p(ry), 20, 21, *outs) == -1        # beyond max_iter
    d2 = _lib.dqp_mpc_dims(B, 9, 9, T, 1, 0)                                         # n + m > 16: no stage-wise kernel
    assert lib.dqp_mpc_qp_stepped_workspace_bytes(ctypes.byref(d2)) == 0


def test_stagewise_flag_agrees_with_nullspace_kernels():
    from diff_qp_mpc_amd import _lib
    lib = _lib.load()
    n, m, T, B = 3, 3, 5, 64
    data = problem(n, m, T, B, seed=3)
    a = _forward(lib, _lib, n, m, T, data, _lib.DQP_FLAG_BATCH_TERMINATION)
    b = _forward(lib, _lib, n, m, T, data, _lib.DQP_FLAG_BATCH_TERMINATION | _lib.DQP_FLAG_STAGEWISE)
    np.testing.assert_allclose(b["tau"].cpu().numpy(), a["tau"].cpu().numpy(), rtol=1e-6, atol=1e-8)


def test_trace_records_every_launch():
    from diff_qp_mpc_amd import _lib
    import diff_qp_mpc_amd as dqp
    from test_gpu_parity import family_R
    ins = [dev(a, grad=True) for a in family_R(0, 256, 30, 30, 15)]
    with _lib.trace(64) as tr:
        z = dqp.QPFunction(verbose=-1)(*ins)
        z.sum().backward()
        torch.cuda.synchronize()
    names = [k for k, _ in tr.records]
    assert any("r16n::forward_kernel" in k and "30, 30, 15" in k for k in names), names
    assert any("r16n::backward_kernel" in k for k in names), names
    assert any("term_scan_kernel" in k for k in names), names
    assert all(0.0 < ms < 100.0 for _, ms in tr.records), tr.records
    by = tr.by_kernel()
    assert sum(c for c, _ in by.values()) == len(tr.records)
    # tracing is off again: a second block starts empty
    with _lib.trace(8) as tr2:
        pass
    assert tr2.records == []
