"""QPFunction / DenseQPFunction with the reference's signatures (qpth/qp.py:19-183, 187-271),
computed by the fused HIP kernels behind the C ABI (include/dqp.h).

Differences from the reference, all documented in DESIGN.md:
  * `dyn_res` / `cost_grad` are accepted positionally (qp.py:24).  The fused kernel evaluates
    cost_grad(x) = Q x + p and, for dyn_res, either the linear form A x - b or -- when dyn_res is a
    dynamics.DynamicsResidual over a registered device model -- the TRUE-dynamics residual on chip,
    every iteration, as the reference does with its Python closure (qp_wrapper.py:309,316).  Any
    other closure is checked against the linear forms at one random point (check_callables=True,
    the default) and refused if it differs: nothing is substituted silently.
  * termination: the reference's batch-coupled rule (batch.py:119-144) is reproduced on the
    device by default (TERMINATION = "batch"); "per_problem" lets every problem stop on its own
    (include/dqp.h) -- faster in very large batches, float-tolerance parity.
  * `check_Q_spd` uses the kernel's Cholesky status instead of B host-side eig calls, and -- like the INACC_ERR
    warning -- reaches the host lazily (CHECKS below): no device synchronisation inside forward.
  * solver=QPSolvers.CVXPY is not available (cvxpy is an offline oracle in the reference).
"""
import ctypes
from enum import Enum

import torch
from torch.autograd import Function

from . import _lib
from .util import expandParam, extract_nBatch

INACC_ERR = """
--------
qpth warning: Returning an inaccurate and potentially incorrect solution.

Some residual is large.
Your problem may be infeasible or difficult.
--------
"""


# per-problem early exit only once the best residual is below this (include/dqp.h)
STALL_TOL = 1e-10
# "batch": DQP_FLAG_BATCH_TERMINATION, the reference's stopping rule replayed over the batch (the
# default of every operator in this package); "per_problem": each problem stops on its own
TERMINATION = "batch"
# shards of one logical batch: a callable int64[3] -> int64[3] that ORs the batch rule's iteration masks over
# the shards (sharding.global_batch_rule sets it); None: the rule couples the batch this call is given
MASK_EXCHANGE = None
# extra dqp_opts.flags OR-ed into every call (tests use DQP_FLAG_GENERIC_ONLY / _NO_NULLSPACE to
# pin a kernel family; 0 = automatic dispatch)
FORCE_FLAGS = 0


# How QPFunction's result checks reach the host (`check_Q_spd` -> RuntimeError('Q is not SPD.'), qp.py:86; the INACC_ERR
# warning for verbose >= 0, batch.py:142-143).  The reference runs them inside forward, which costs a device
# synchronisation per solve.  "lazy" (default): forward only enqueues a two-flag reduction and its copy to pinned host
# memory; the flags are looked at -- and the error raised / the warning printed -- at the first of: the backward of that
# call, the next forward of any QPFunction (once the copy has landed), or flush_checks().  "sync": as the reference.
CHECKS = "lazy"
_pending = []


def _enqueue_checks(info, resid, want_spd, want_inacc):
    flags = torch.stack(((info[:, 0] == _lib.DQP_STATUS_Q_NOT_PD).any(), resid.max() > 1.0))
    host = torch.empty(2, dtype=torch.bool, pin_memory=True)
    host.copy_(flags, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(info.device))
    entry = [ev, host, want_spd, want_inacc]
    _pending.append(entry)
    return entry


def _resolve(entry, wait):
    """Look at one pending check (waiting for its copy if asked to); returns False while it is still in flight."""
    ev, host, want_spd, want_inacc = entry
    if not wait and not ev.query():
        return False
    ev.synchronize()
    if entry in _pending:
        _pending.remove(entry)
    if want_inacc and bool(host[1]):
        print(INACC_ERR)                                             # batch.py:142-143
    if want_spd and bool(host[0]):
        raise RuntimeError('Q is not SPD.')                          # qp.py:86
    return True


def flush_checks(wait=True):
    """Resolve the lazy result checks of earlier forward calls (raises / prints as the reference's forward would have)."""
    for entry in list(_pending):
        _resolve(entry, wait)


class QPSolvers(Enum):
    PDIPM_BATCHED = 1
    CVXPY = 2


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else ctypes.c_void_p(0)


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _prep(t, nd):
    """-> (contiguous fp64 tensor, batch stride in elements; 0 when shared)."""
    t = t.detach()
    if t.dtype != torch.float64:
        t = t.double()
    t = t.contiguous()
    if t.numel() == 0:
        return t, 0
    if t.dim() == nd:
        return t, t[0].numel()
    if t.dim() == nd - 1:
        return t, 0
    raise RuntimeError("Unexpected number of dimensions.")


def _require_gpu(*ts):
    for t in ts:
        if t is not None and t.numel() > 0 and not t.is_cuda:
            raise RuntimeError(
                "diff_qp_mpc_amd operators run only on a GPU (HIP); got a %s tensor. "
                "There is no CPU fallback." % t.device)


def _forward_impl(Q_, p_, G_, h_, A_, b_, eps, maxIter, notImprovedLim, termination=None, dyn=None):
    _require_gpu(Q_, p_, G_, h_, A_, b_)
    termination = termination or TERMINATION
    if termination not in ("batch", "per_problem"):
        raise ValueError("termination must be 'batch' or 'per_problem'")
    lib = _lib.load()
    nBatch = extract_nBatch(Q_, p_, G_, h_, A_, b_)
    Q, sQ = _prep(Q_, 3)
    p, sp = _prep(p_, 2)
    G, sG = _prep(G_, 3)
    h, sh = _prep(h_, 2)
    A, sA = _prep(A_, 3)
    b, sb = _prep(b_, 2)
    nineq, nz = G.shape[-2], G.shape[-1]
    neq = A.shape[-2] if A.numel() > 0 else 0
    assert neq > 0 or nineq > 0                                   # qp.py:90
    dev = Q.device
    dims = _lib.dqp_dims(nBatch, nz, nineq, neq, sQ, sp, sG, sh, sA, sb)
    if termination == "batch" and not (1 <= maxIter <= 64):
        raise ValueError("the batch-coupled termination replays the stop rule on 64-bit iteration masks: 1 <= maxIter <= 64 "
                         "(got %d); use termination='per_problem' for longer runs" % maxIter)
    flags = FORCE_FLAGS | (_lib.DQP_FLAG_BATCH_TERMINATION if termination == "batch" else 0)
    opts = _lib.dqp_opts(eps, STALL_TOL, maxIter, notImprovedLim, flags, 0)
    if dyn is not None:          # true-dynamics residual on chip (include/dqp.h: dqp_opts.dyn_*)
        if dyn.x0.shape[0] != nBatch or dyn.x0.device != dev:
            raise RuntimeError("DynamicsResidual.x0 must be (nBatch, n_state) on the QP's device")
        opts.dyn_id, opts.dyn_T, opts.dyn_dt = dyn.dynamics.id, dyn.T, dyn.dynamics.dt
        opts.dyn_x0 = dyn.x0.data_ptr()
    kw = dict(dtype=torch.float64, device=dev)
    zhat = torch.empty(nBatch, nz, **kw)
    lam = torch.empty(nBatch, nineq, **kw)
    slack = torch.empty(nBatch, nineq, **kw)
    nu = torch.empty(nBatch, neq, **kw)
    info = torch.empty(nBatch, 2, dtype=torch.int32, device=dev)
    resid = torch.empty(nBatch, **kw)
    # scratch for the null-space forward kernels (include/dqp.h: dqp_workspace_bytes)
    wsb = int(lib.dqp_workspace_bytes(ctypes.byref(dims)))
    ws = torch.empty(wsb // 8, **kw) if wsb > 0 else None
    tb = int(lib.dqp_termination_bytes(ctypes.byref(dims), ctypes.byref(opts)))
    term = torch.empty((tb + 7) // 8, **kw) if tb > 0 else None
    args = (_ptr(Q), _ptr(p), _ptr(G), _ptr(h), _ptr(A), _ptr(b), _ptr(zhat), _ptr(lam), _ptr(nu), _ptr(slack),
            _ptr(info), _ptr(resid), _ptr(ws), _ptr(term))
    if termination == "batch" and MASK_EXCHANGE is not None:
        # this batch is a shard of a larger one (sharding.global_batch_rule): pass 1, this shard's three
        # iteration masks, the caller's OR over the shards (24 bytes), the rule on the combined masks, pass 2
        opts1 = _lib.dqp_opts(eps, STALL_TOL, maxIter, notImprovedLim, flags | _lib.DQP_FLAG_HISTORY_ONLY, 0)
        if dyn is not None:
            opts1.dyn_id, opts1.dyn_T, opts1.dyn_dt, opts1.dyn_x0 = opts.dyn_id, opts.dyn_T, opts.dyn_dt, opts.dyn_x0
        masks = torch.zeros(3, dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            rc = lib.dqp_qp_forward(ctypes.byref(dims), ctypes.byref(opts1), *args, _stream(dev))
            _lib.check(rc, "dqp_qp_forward (pass 1)")
            _lib.check(lib.dqp_term_local_masks(ctypes.byref(dims), ctypes.byref(opts), _ptr(term), _ptr(masks),
                                                _stream(dev)), "dqp_term_local_masks")
        masks = MASK_EXCHANGE(masks).contiguous()
        with torch.cuda.device(dev):
            rc = lib.dqp_qp_forward_finish(ctypes.byref(dims), ctypes.byref(opts), *args, _ptr(masks), _stream(dev))
        _lib.check(rc, "dqp_qp_forward_finish")
    else:
        with torch.cuda.device(dev):
            rc = lib.dqp_qp_forward(ctypes.byref(dims), ctypes.byref(opts), *args, _stream(dev))
        _lib.check(rc, "dqp_qp_forward")
    # the workspace now holds the factorisation context backward can restart from (include/dqp.h)
    # (only the null-space kernels leave one: not the forced families, not the true-dynamics path)
    big = max(nz, nineq, neq) > _lib.DQP_MAX_DIM         # one QP per workgroup, matrices in the workspace (dqp_big.hip)
    ctx_ws = ws if (ws is not None and dyn is None and
                    (big or not (FORCE_FLAGS & (_lib.DQP_FLAG_NO_NULLSPACE | _lib.DQP_FLAG_GENERIC_ONLY)))) else None
    return zhat, lam, nu, slack, info, resid, (Q, G, A, dims, ctx_ws)


def _backward_impl(saved, zhat, lam, nu, slack, dl_dzhat, need, flags):
    lib = _lib.load()
    Q, G, A, dims, ctx_ws = saved
    nBatch, nz, nineq, neq = dims.nbatch, dims.nz, dims.nineq, dims.neq
    dev = Q.device
    kw = dict(dtype=torch.float64, device=dev)
    g = dl_dzhat.detach().double().contiguous()
    dQ = torch.empty(nBatch, nz, nz, **kw) if need[0] else None
    dp = torch.empty(nBatch, nz, **kw) if need[1] else None
    dG = torch.empty(nBatch, nineq, nz, **kw) if need[2] else None
    dh = torch.empty(nBatch, nineq, **kw) if need[3] else None
    dA = torch.empty(nBatch, neq, nz, **kw) if (need[4] and neq > 0) else None
    db = torch.empty(nBatch, neq, **kw) if (need[5] and neq > 0) else None
    big = max(nz, nineq, neq) > _lib.DQP_MAX_DIM
    if ctx_ws is not None and (big or not (FORCE_FLAGS & (_lib.DQP_FLAG_NO_NULLSPACE | _lib.DQP_FLAG_GENERIC_ONLY))):
        flags |= _lib.DQP_FLAG_BACKWARD_CTX
    opts = _lib.dqp_opts(0.0, 0.0, 0, 0, flags | FORCE_FLAGS, 0)
    with torch.cuda.device(dev):
        rc = lib.dqp_qp_backward(ctypes.byref(dims), ctypes.byref(opts), _ptr(Q), _ptr(G), _ptr(A),
                                 _ptr(zhat), _ptr(lam), _ptr(nu), _ptr(slack), _ptr(g),
                                 _ptr(dQ), _ptr(dp), _ptr(dG), _ptr(dh), _ptr(dA), _ptr(db),
                                 ctypes.c_void_p(0), _ptr(ctx_ws), _stream(dev))
    _lib.check(rc, "dqp_qp_backward")
    return dQ, dp, dG, dh, dA, db


def _check_callables(Q, p, A, b, dyn_res, cost_grad, nBatch):
    Qe, _ = expandParam(Q, nBatch, 3)
    x = torch.randn(nBatch, Qe.shape[-1], dtype=Qe.dtype, device=Qe.device)
    if cost_grad is not None:
        want = torch.bmm(Qe, x.unsqueeze(-1)).squeeze(-1) + p
        if not torch.allclose(cost_grad(x), want, rtol=1e-8, atol=1e-10):
            raise RuntimeError("cost_grad(x) != Qx + p: only the linear form is fused on chip")
    if dyn_res is not None and A.numel() > 0:
        Ae, _ = expandParam(A, nBatch, 3)
        want = torch.bmm(Ae, x.unsqueeze(-1)).squeeze(-1) - b
        if not torch.allclose(dyn_res(x), want, rtol=1e-8, atol=1e-10):
            raise RuntimeError("dyn_res(x) != Ax - b: a nonlinear residual is fused on chip only for "
                               "registered device models (pass a dynamics.DynamicsResidual)")


def _split_dyn_res(dyn_res):
    """-> (DynamicsResidual for the fused path or None, closure that still needs the linearity check)"""
    from .dynamics import DynamicsResidual
    if isinstance(dyn_res, DynamicsResidual):
        return dyn_res, None
    return None, dyn_res


def QPFunction(eps=1e-12, verbose=0, notImprovedLim=3, maxIter=20,
               solver=QPSolvers.PDIPM_BATCHED, check_Q_spd=True, check_callables=True):
    """Factory with the reference's signature (qpth/qp.py:19-21); returns a callable
    `(Q, p, G, h, A, b, dyn_res=None, cost_grad=None) -> zhat (nBatch, nz)`."""
    if solver != QPSolvers.PDIPM_BATCHED:
        raise NotImplementedError("only QPSolvers.PDIPM_BATCHED is implemented on MI355X")

    class QPFunctionFn(Function):
        @staticmethod
        def forward(ctx, Q_, p_, G_, h_, A_, b_, dyn_res=None, cost_grad=None):
            nBatch = extract_nBatch(Q_, p_, G_, h_, A_, b_)
            dyn, closure = _split_dyn_res(dyn_res)
            if check_callables and (closure is not None or cost_grad is not None):
                _check_callables(Q_, p_, A_, b_, closure, cost_grad, nBatch)
            zhat, lam, nu, slack, info, resid, saved = _forward_impl(
                Q_, p_, G_, h_, A_, b_, eps, maxIter, notImprovedLim, dyn=dyn)
            ctx.check = None
            if check_Q_spd or verbose >= 0:
                flush_checks(wait=False)                             # earlier calls whose flags have landed
                ctx.check = _enqueue_checks(info, resid, check_Q_spd, verbose >= 0)
                if CHECKS == "sync":
                    _resolve(ctx.check, True)
            ctx.saved = saved
            ctx.lams, ctx.nus, ctx.slacks, ctx.info = lam, nu, slack, info
            ctx.shared = tuple(t.numel() > 0 and t.dim() == nd - 1 for t, nd in
                               zip((Q_, p_, G_, h_, A_, b_), (3, 2, 3, 2, 3, 2)))
            ctx.neq = saved[3].neq
            ctx.out_dtype = Q_.dtype
            ctx.save_for_backward(zhat)
            return zhat.to(Q_.dtype)

        @staticmethod
        def backward(ctx, dl_dzhat):
            zhat, = ctx.saved_tensors
            if ctx.check is not None and ctx.check in _pending:
                _resolve(ctx.check, True)
            need = ctx.needs_input_grad[:6]
            grads = list(_backward_impl(ctx.saved, zhat, ctx.lams, ctx.nus, ctx.slacks,
                                        dl_dzhat, need, 0))
            for i, g in enumerate(grads):
                if g is None:
                    continue
                if ctx.shared[i]:
                    g = g.mean(0)                                    # qp.py:160-178
                grads[i] = g.to(ctx.out_dtype)
            return tuple(grads) + (None, None)

    def apply(Q, p, G, h, A, b, dyn_res=None, cost_grad=None):
        return QPFunctionFn.apply(Q, p, G, h, A, b, dyn_res, cost_grad)

    return apply


def DenseQPFunction(bsz=1, eps=1e-12, verbose=0, notImprovedLim=3, maxIter=20, check_callables=True):
    """Factory with the reference's signature (qpth/qp.py:187-188); returns a callable
    `(Q, p, G, h, A, b, dyn_res, cost_grad=None) -> zhat`.  All six parameters must be
    batched (the reference's preprocess(), qp.py:195-217, does no expandParam).

    The reference solves the same Newton systems through a regularised (1e-7) full-KKT LU
    with one refinement step (batch_LU.py:212-244); here they are solved by the fused
    block-Cholesky kernel, and backward uses d = lam/slack without QPFunction's clamps
    (qp.py:246-250)."""

    class Solver(Function):
        @staticmethod
        def forward(ctx, Q, p, G, h, A, b, dyn_res=None, cost_grad=None):
            for t, nd in zip((Q, p, G, h, A, b), (3, 2, 3, 2, 3, 2)):
                if t.dim() != nd:
                    raise RuntimeError("DenseQPFunction needs batched parameters")
            dyn, closure = _split_dyn_res(dyn_res)
            if check_callables and (closure is not None or cost_grad is not None):
                _check_callables(Q, p, A, b, closure, cost_grad, Q.shape[0])
            zhat, lam, nu, slack, info, resid, saved = _forward_impl(
                Q, p, G, h, A, b, eps, maxIter, notImprovedLim, dyn=dyn)
            ctx.saved = saved
            ctx.lams, ctx.nus, ctx.slacks, ctx.info = lam, nu, slack, info
            ctx.out_dtype = Q.dtype
            ctx.save_for_backward(zhat)
            return zhat.to(Q.dtype)

        @staticmethod
        def backward(ctx, dl_dzhat):
            zhat, = ctx.saved_tensors
            need = ctx.needs_input_grad[:6]
            grads = _backward_impl(ctx.saved, zhat, ctx.lams, ctx.nus, ctx.slacks, dl_dzhat,
                                   need, _lib.DQP_FLAG_DENSE_BACKWARD)
            grads = tuple(None if g is None else g.to(ctx.out_dtype) for g in grads)
            return grads + (None, None)

    def apply(Q, p, G, h, A, b, dyn_res=None, cost_grad=None):
        return Solver.apply(Q, p, G, h, A, b, dyn_res, cost_grad)

    return apply
