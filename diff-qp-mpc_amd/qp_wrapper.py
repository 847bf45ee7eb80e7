"""qp_wrapper.MPC with the reference's interface (qpth/qp_wrapper.py:22-34, 59-692).

    MPC(n_state, n_ctrl, T, u_lower=..., u_upper=..., qp_iter=..., single_qp_solve=..., ...)
        (x0, QuadCost(C, c), LinDx(F, f) | dynamics callable, dx_jac[, dx_true]) -> (x, u)

time-major tensors, as in the reference: C (T,B,nt,nt) c (T,B,nt) F (T-1,B,n,nt) f (T-1,B,n)
x0 (B,n); returns x (T,B,n), u (T,B,m).

What runs where:
  * dense QP assembly (compute_Qq/Ab/Gh_dense, qp_wrapper.py:638-679) and its adjoint: HIP
    kernels behind dqp_mpc_assemble / dqp_mpc_assemble_backward (csrc/dqp_mpc.hip);
  * the QP itself (qp.DenseQPFunction, qp_wrapper.py:316): fused HIP PDIPM (csrc/dqp_*.hip);
  * the SQP outer loop, rollout, cost and line search (behaviour of qp_wrapper.py:298-436,
    598-611, 690-692): vectorised torch ops on the same device (they call user-supplied Python
    dynamics, which cannot be fused).

The reference passes the TRUE-dynamics residual closure into the PDIPM (`dyn_res_lam`,
qp_wrapper.py:309,316).  Here: for LinDx that residual IS A z - b (what the kernels evaluate); for
a dynamics.DeviceDynamics (registered device model) the fused PDIPM evaluates the true residual
on chip every iteration, like the reference; any other dynamics module is refused unless the
caller opts into the linearised residual (`linearised_residual=True`, an extension) -- nothing is
substituted silently.
"""
import ctypes
from enum import Enum
from typing import NamedTuple, Optional

import torch
from torch.autograd import Function
from torch.nn import Module

from . import _lib
from .dynamics import DeviceDynamics, DynamicsResidual
from .qp import DenseQPFunction


class QuadCost(NamedTuple):
    """Quadratic stage cost 1/2 tau^T C tau + c^T tau (same fields as the reference's type)."""
    C: Optional[torch.Tensor] = None
    c: Optional[torch.Tensor] = None


class LinDx(NamedTuple):
    """Affine dynamics x_{t+1} = F_t [x_t; u_t] + f_t (same fields as the reference's type)."""
    F: Optional[torch.Tensor] = None
    f: Optional[torch.Tensor] = None


# single_qp goes through the fused MPC QP entry points (dqp_mpc_qp_*) when the size has a null-space
# kernel; False forces the assemble + DenseQPFunction pipeline (tests compare the two)
FUSED_MPC_QP = True
# line_search (rollouts + costs + backtracking) as one kernel for LinDx / registered device models
FUSED_LINE_SEARCH = True

# member names and values as in qpth.qp_wrapper.GradMethods
GradMethods = Enum("GradMethods", ["AUTO_DIFF", "FINITE_DIFF", "ANALYTIC", "ANALYTIC_CHECK"])


def detach_maybe(x):
    """None stays None; tensors that track gradients are detached."""
    if x is None or not x.requires_grad:
        return x
    return x.detach()


def bmv(X, y):
    """Batched matrix-vector product (B,r,c) x (B,c) -> (B,r)."""
    return torch.matmul(X, y[..., None])[..., 0]


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


# OR-ed into dqp_opts.flags of dqp_mpc_qp_forward (tests: DQP_FLAG_RIC_GLOBAL_WS pins the stage-wise kernels'
# workspace to the caller's buffer where the default keeps it in LDS)
EXTRA_FLAGS = 0


class _AssembleDenseQP(Function):
    """(C, c, F, f, x0) -> (Q, p, G, h, A, b) on the device (qp_wrapper.py:638-679)."""

    @staticmethod
    def forward(ctx, C, c, F, f, x0, u_lower, u_upper, n_state, n_ctrl, T):
        lib = _lib.load()
        for t in (C, c, F, f, x0):
            if not t.is_cuda:
                raise RuntimeError("diff_qp_mpc_amd.qp_wrapper.MPC runs only on a GPU (HIP); "
                                   "there is no CPU fallback.")
        dev = x0.device
        B = x0.shape[0]
        nt = n_state + n_ctrl
        nz, neq = T * nt, T * n_state
        bounds = u_upper is not None
        nineq = 2 * T * n_ctrl if bounds else n_ctrl
        cv = lambda t: t.detach().double().contiguous()
        Cc, cc, Fc, fc, xc = cv(C), cv(c), cv(F), cv(f), cv(x0)
        ul = cv(u_lower).reshape(-1) if bounds else None
        uu = cv(u_upper).reshape(-1) if bounds else None
        if bounds and (ul.numel() != n_ctrl or uu.numel() != n_ctrl):
            raise RuntimeError("u_lower/u_upper must have shape (n_ctrl,) (qp_wrapper.py:677-678)")
        kw = dict(dtype=torch.float64, device=dev)
        Q = torch.empty(B, nz, nz, **kw); p = torch.empty(B, nz, **kw)
        G = torch.empty(B, nineq, nz, **kw); h = torch.empty(B, nineq, **kw)
        A = torch.empty(B, neq, nz, **kw); b = torch.empty(B, neq, **kw)
        dims = _lib.dqp_mpc_dims(B, n_state, n_ctrl, T, 1 if bounds else 0, 0)
        with torch.cuda.device(dev):
            rc = lib.dqp_mpc_assemble(ctypes.byref(dims), _ptr(Cc), _ptr(cc), _ptr(Fc), _ptr(fc),
                                      _ptr(xc), _ptr(ul), _ptr(uu), _ptr(Q), _ptr(p), _ptr(G),
                                      _ptr(h), _ptr(A), _ptr(b), _stream(dev))
        _lib.check(rc, "dqp_mpc_assemble")
        ctx.dims = dims
        ctx.shapes = (C.shape, c.shape, F.shape, f.shape, x0.shape)
        ctx.dtype = x0.dtype
        ctx.mark_non_differentiable(G, h)
        return Q, p, G, h, A, b

    @staticmethod
    def backward(ctx, dQ, dp, dG, dh, dA, db):
        lib = _lib.load()
        dims = ctx.dims
        dev = next(t for t in (dQ, dp, dA, db) if t is not None).device
        kw = dict(dtype=torch.float64, device=dev)
        need = ctx.needs_input_grad
        outs = [torch.empty(s, **kw) if need[i] else None for i, s in enumerate(ctx.shapes)]
        cv = lambda t: None if t is None else t.double().contiguous()
        dQ, dp, dA, db = cv(dQ), cv(dp), cv(dA), cv(db)
        with torch.cuda.device(dev):
            rc = lib.dqp_mpc_assemble_backward(ctypes.byref(dims), _ptr(dQ), _ptr(dp), _ptr(dA),
                                               _ptr(db), *[_ptr(o) for o in outs], _stream(dev))
        _lib.check(rc, "dqp_mpc_assemble_backward")
        outs = [None if o is None else o.to(ctx.dtype) for o in outs]
        return (*outs, None, None, None, None, None)


class _MPCQP(Function):
    """(C, c, F, f, x0) -> tau (B, T, n+m): assembly + DenseQPFunction (qp_wrapper.py:311-319) as ONE
    solve through dqp_mpc_qp_forward -- the dense (Q,p,G,h,A,b) exists only in registers -- and
    the backward straight into (dC, dc, dF, df, dx0) (dqp_mpc_qp_backward)."""

    @staticmethod
    def supported(B, n_state, n_ctrl, T, dyn=None):
        dims = _lib.dqp_mpc_dims(B, n_state, n_ctrl, T, 1, dyn.id if dyn is not None else 0)
        return bool(_lib.load().dqp_mpc_qp_supported(ctypes.byref(dims)))

    @staticmethod
    def forward(ctx, C, c, F, f, x0, u_lower, u_upper, n_state, n_ctrl, T, dyn=None):
        from . import qp as qpmod
        lib = _lib.load()
        for t in (C, c, F, f, x0):
            if not t.is_cuda:
                raise RuntimeError("diff_qp_mpc_amd.qp_wrapper.MPC runs only on a GPU (HIP); "
                                   "there is no CPU fallback.")
        dev, B, nt = x0.device, x0.shape[0], n_state + n_ctrl
        cv = lambda t: t.detach().double().contiguous()
        keep = [cv(C), cv(c), cv(F), cv(f), cv(x0), cv(u_lower).reshape(-1), cv(u_upper).reshape(-1)]
        if keep[5].numel() != n_ctrl or keep[6].numel() != n_ctrl:
            raise RuntimeError("u_lower/u_upper must have shape (n_ctrl,) (qp_wrapper.py:677-678)")
        # dyn: a DeviceDynamics whose true step is the equality residual of the iterations (the reference's
        # dyn_res closure, qp_wrapper.py:309,316); F, f stay the linearisation the Newton steps use
        dims = _lib.dqp_mpc_dims(B, n_state, n_ctrl, T, 1, dyn.id if dyn is not None else 0)
        batch = qpmod.TERMINATION == "batch"
        opts = _lib.dqp_opts(1e-12, qpmod.STALL_TOL, 20, 3, (_lib.DQP_FLAG_BATCH_TERMINATION if batch else 0) | EXTRA_FLAGS, 0)
        if dyn is not None:
            opts.dyn_dt = dyn.dt
        kw = dict(dtype=torch.float64, device=dev)
        tau = torch.empty(B, T, nt, **kw)
        lam = torch.empty(B, 2 * T * n_ctrl, **kw); slack = torch.empty(B, 2 * T * n_ctrl, **kw)
        nu = torch.empty(B, T * n_state, **kw)
        info = torch.empty(B, 2, dtype=torch.int32, device=dev)
        resid = torch.empty(B, **kw)
        ws = torch.empty(int(lib.dqp_mpc_qp_workspace_bytes(ctypes.byref(dims))) // 8, **kw)
        tb = int(lib.dqp_mpc_qp_termination_bytes(ctypes.byref(dims), ctypes.byref(opts)))
        term = torch.empty((tb + 7) // 8, **kw) if tb else None
        with torch.cuda.device(dev):
            rc = lib.dqp_mpc_qp_forward(ctypes.byref(dims), ctypes.byref(opts), *[_ptr(t) for t in keep],
                                        _ptr(tau), _ptr(lam), _ptr(nu), _ptr(slack), _ptr(info), _ptr(resid),
                                        _ptr(ws), _ptr(term), _stream(dev))
        _lib.check(rc, "dqp_mpc_qp_forward")
        ctx.dims, ctx.ws = dims, ws
        ctx.shapes = (C.shape, c.shape, F.shape, f.shape, x0.shape)
        ctx.dtype = x0.dtype
        ctx.save_for_backward(tau, lam, nu, slack, keep[0], keep[2])     # C, F: the stage-wise backward refactors
        ctx.info, ctx.resid = info, resid
        return tau.to(x0.dtype)

    @staticmethod
    def backward(ctx, dtau):
        lib = _lib.load()
        tau, lam, nu, slack, C64, F64 = ctx.saved_tensors
        dev = tau.device
        kw = dict(dtype=torch.float64, device=dev)
        need = ctx.needs_input_grad
        outs = [torch.empty(s, **kw) if need[i] else None for i, s in enumerate(ctx.shapes)]
        g = dtau.detach().double().contiguous()
        opts = _lib.dqp_opts(0.0, 0.0, 0, 0, _lib.DQP_FLAG_DENSE_BACKWARD, 0)
        with torch.cuda.device(dev):
            rc = lib.dqp_mpc_qp_backward(ctypes.byref(ctx.dims), ctypes.byref(opts), _ptr(C64), _ptr(F64), _ptr(tau), _ptr(lam),
                                         _ptr(nu), _ptr(slack), _ptr(g), *[_ptr(o) for o in outs],
                                         ctypes.c_void_p(0), _ptr(ctx.ws), _stream(dev))
        _lib.check(rc, "dqp_mpc_qp_backward")
        outs = [None if o is None else o.to(ctx.dtype) for o in outs]
        return (*outs, None, None, None, None, None, None)


class _MPCQPStepped(Function):
    """_MPCQP for a dynamics model the library cannot evaluate (a caller's torch module): the reference calls its
    dyn_res closure once per PDIPM iteration on the current iterate (qp_wrapper.py:309,316 -> batch_LU.py:97).
    Here: one stage-wise PDIPM iteration per C-ABI call (dqp_mpc_qp_forward_stepped), the module evaluated in
    between on the iterate the call hands back -- one torch evaluation per iteration, everything else on chip."""

    @staticmethod
    def supported(B, n_state, n_ctrl, T):
        dims = _lib.dqp_mpc_dims(B, n_state, n_ctrl, T, 1, 0)
        return int(_lib.load().dqp_mpc_qp_stepped_workspace_bytes(ctypes.byref(dims))) > 0

    @staticmethod
    def forward(ctx, C, c, F, f, x0, u_lower, u_upper, n_state, n_ctrl, T, residual):
        from . import qp as qpmod
        lib = _lib.load()
        dev, B, nt = x0.device, x0.shape[0], n_state + n_ctrl
        cv = lambda t: t.detach().double().contiguous()
        keep = [cv(C), cv(c), cv(F), cv(f), cv(x0), cv(u_lower).reshape(-1), cv(u_upper).reshape(-1)]
        if keep[5].numel() != n_ctrl or keep[6].numel() != n_ctrl:
            raise RuntimeError("u_lower/u_upper must have shape (n_ctrl,) (qp_wrapper.py:677-678)")
        dims = _lib.dqp_mpc_dims(B, n_state, n_ctrl, T, 1, 0)
        batch = qpmod.TERMINATION == "batch"
        max_iter = 20
        opts = _lib.dqp_opts(1e-12, qpmod.STALL_TOL, max_iter, 3, (_lib.DQP_FLAG_BATCH_TERMINATION if batch else 0) | EXTRA_FLAGS, 0)
        kw = dict(dtype=torch.float64, device=dev)
        tau = torch.empty(B, T, nt, **kw)
        lam = torch.empty(B, 2 * T * n_ctrl, **kw); slack = torch.empty(B, 2 * T * n_ctrl, **kw)
        nu = torch.empty(B, T * n_state, **kw)
        info = torch.empty(B, 2, dtype=torch.int32, device=dev)
        resid = torch.empty(B, **kw)
        ws = torch.empty(int(lib.dqp_mpc_qp_stepped_workspace_bytes(ctypes.byref(dims))) // 8, **kw)
        tb = int(lib.dqp_mpc_qp_stepped_termination_bytes(ctypes.byref(dims), ctypes.byref(opts)))
        term = torch.empty((tb + 7) // 8, **kw) if tb else None

        def call(ry, it0, it1):
            with torch.cuda.device(dev):
                rc = lib.dqp_mpc_qp_forward_stepped(ctypes.byref(dims), ctypes.byref(opts), *[_ptr(t) for t in keep], _ptr(ry),
                                                    it0, it1, _ptr(tau), _ptr(lam), _ptr(nu), _ptr(slack), _ptr(info),
                                                    _ptr(resid), _ptr(ws), _ptr(term), _stream(dev))
            _lib.check(rc, "dqp_mpc_qp_forward_stepped")

        with torch.no_grad():
            call(None, 0, 0)                                    # starting point -> tau
            for it in range(max_iter):
                ry = residual(tau.reshape(B, -1).to(x0.dtype)).detach().double().contiguous()
                call(ry, it, it + 1)                            # the last call leaves the best iterate in tau
        ctx.dims, ctx.ws = dims, ws
        ctx.shapes = (C.shape, c.shape, F.shape, f.shape, x0.shape)
        ctx.dtype = x0.dtype
        ctx.save_for_backward(tau, lam, nu, slack, keep[0], keep[2])
        ctx.info, ctx.resid = info, resid
        return tau.to(x0.dtype)

    @staticmethod
    def backward(ctx, dtau):
        lib = _lib.load()
        tau, lam, nu, slack, C64, F64 = ctx.saved_tensors
        dev = tau.device
        kw = dict(dtype=torch.float64, device=dev)
        need = ctx.needs_input_grad
        outs = [torch.empty(s, **kw) if need[i] else None for i, s in enumerate(ctx.shapes)]
        g = dtau.detach().double().contiguous()
        opts = _lib.dqp_opts(0.0, 0.0, 0, 0, _lib.DQP_FLAG_DENSE_BACKWARD | _lib.DQP_FLAG_STAGEWISE, 0)
        with torch.cuda.device(dev):
            rc = lib.dqp_mpc_qp_backward(ctypes.byref(ctx.dims), ctypes.byref(opts), _ptr(C64), _ptr(F64), _ptr(tau), _ptr(lam),
                                         _ptr(nu), _ptr(slack), _ptr(g), *[_ptr(o) for o in outs],
                                         ctypes.c_void_p(0), _ptr(ctx.ws), _stream(dev))
        _lib.check(rc, "dqp_mpc_qp_backward")
        outs = [None if o is None else o.to(ctx.dtype) for o in outs]
        return (*outs, None, None, None, None, None, None)


class _Rollout(Function):
    """rollout(x0, u, dynamics) as one launch each way (dqp_mpc_line_search with C == NULL;
    dqp_mpc_rollout_backward)."""

    @staticmethod
    def forward(ctx, x0, u, F, f, dyn, n, m, T):
        lib = _lib.load()
        dev, B = x0.device, x0.shape[0]
        cv = lambda t: t.detach().double().contiguous()
        lin = dyn is None
        keep = [cv(F) if lin else None, cv(f) if lin else None, cv(x0), None, cv(u), None, None, None]
        kw = dict(dtype=torch.float64, device=dev)
        x_new, u_new = torch.empty(T, B, n, **kw), torch.empty(T, B, m, **kw)
        alpha, cost_new = torch.empty(B, **kw), torch.empty(B, **kw)
        dims = _lib.dqp_mpc_dims(B, n, m, T, 1, 0)
        with torch.cuda.device(dev):
            rc = lib.dqp_mpc_line_search(ctypes.byref(dims), 0 if lin else dyn.id, 0.0 if lin else dyn.dt,
                                         *[_ptr(t) for t in keep], 1.0, 1, _ptr(x_new), _ptr(u_new), _ptr(alpha),
                                         _ptr(cost_new), _stream(dev))
        _lib.check(rc, "dqp_mpc_line_search")
        ctx.dims, ctx.dyn, ctx.dtype = dims, dyn, x0.dtype
        ctx.shapes = (F.shape if lin else None, f.shape if lin else None)
        ctx.save_for_backward(x_new, u_new, keep[0] if lin else x_new)
        return x_new.to(x0.dtype)

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, u, F = ctx.saved_tensors
        dyn, dims = ctx.dyn, ctx.dims
        lin = dyn is None
        dev = x.device
        kw = dict(dtype=torch.float64, device=dev)
        need = ctx.needs_input_grad
        g64 = g.detach().double().contiguous()
        d_x0 = torch.empty(dims.nbatch, dims.n_state, **kw) if need[0] else None
        d_u = torch.empty(dims.T, dims.nbatch, dims.n_ctrl, **kw) if need[1] else None
        d_F = torch.empty(ctx.shapes[0], **kw) if (lin and need[2]) else None
        d_f = torch.empty(ctx.shapes[1], **kw) if (lin and need[3]) else None
        with torch.cuda.device(dev):
            rc = lib.dqp_mpc_rollout_backward(ctypes.byref(dims), 0 if lin else dyn.id, 0.0 if lin else dyn.dt,
                                              _ptr(F) if lin else ctypes.c_void_p(0), _ptr(x), _ptr(u), _ptr(g64),
                                              _ptr(d_x0), _ptr(d_u), _ptr(d_F), _ptr(d_f), _stream(dev))
        _lib.check(rc, "dqp_mpc_rollout_backward")
        cvt = lambda t: None if t is None else t.to(ctx.dtype)
        return cvt(d_x0), cvt(d_u), cvt(d_F), cvt(d_f), None, None, None, None


class MPC(Module):
    """Differentiable box-constrained MPC via dense QPs (qpth/qp_wrapper.py:59-211).

    Constructor arguments, defaults and attribute names follow the reference
    (qp_wrapper.py:124-150); options it accepts but never reads (delta_u, back_eps,
    exit_unconverged, detach_unconverged, backprop, slew_rate_penalty, prev_ctrl, u_zero_I) are
    accepted and stored likewise.
    """

    def __init__(self, n_state, n_ctrl, T, u_lower=None, u_upper=None, u_zero_I=None,
                 u_init=None, x_init=None, qp_iter=10, grad_method=GradMethods.ANALYTIC,
                 delta_u=None, verbose=0, eps=1e-7, back_eps=1e-7, n_batch=None,
                 linesearch_decay=0.2, max_linesearch_iter=10, exit_unconverged=True,
                 detach_unconverged=True, backprop=True, slew_rate_penalty=None, prev_ctrl=None,
                 not_improved_lim=5, best_cost_eps=1e-4, solver_type='dense',
                 single_qp_solve=False, add_goal_constraint=False, x_goal=None,
                 linearised_residual=False):
        given = dict(locals())
        super().__init__()
        if (u_lower is None) != (u_upper is None):
            raise AssertionError("give both control bounds or neither")
        if max_linesearch_iter <= 0:
            raise AssertionError("max_linesearch_iter must be positive")
        if solver_type != 'dense':
            raise NotImplementedError("only solver_type='dense' exists in the reference too")
        # every option is kept under the reference's attribute name; tensors that arrive with a
        # graph attached are detached (bounds may also be plain floats)
        plain = ("n_state", "n_ctrl", "T", "qp_iter", "grad_method", "delta_u", "verbose", "eps",
                 "back_eps", "n_batch", "linesearch_decay", "max_linesearch_iter", "exit_unconverged",
                 "detach_unconverged", "backprop", "slew_rate_penalty", "prev_ctrl", "not_improved_lim",
                 "best_cost_eps", "solver_type", "single_qp_solve", "add_goal_constraint", "x_goal",
                 "linearised_residual")
        for name in plain:
            setattr(self, name, given[name])
        for name in ("u_lower", "u_upper", "u_zero_I", "u_init", "x_init"):
            v = given[name]
            setattr(self, name, v if isinstance(v, float) else detach_maybe(v))

    # ------------------------------------------------------------------ behaviour of qp_wrapper.py:213-296
    def _time_major(self, t, full_rank, n_batch, what):
        """Broadcast a (…)-shaped cost / initial-guess tensor to its time-major batched form
        (T, B, …): tensors lacking the batch axis, or both the time and batch axes, are expanded."""
        missing = full_rank - t.dim()
        if missing == 2:
            t = t[None, None].expand(self.T, n_batch, *t.shape)
        elif missing == 1:
            t = t[:, None].expand(t.shape[0], n_batch, *t.shape[1:])
        if t.dim() != full_rank:
            raise SystemExit("MPC Error: Unexpected %s shape." % what)
        return t

    def forward(self, x0, cost, dx, dx_jac, dx_true=None):
        self.dx_true = dx_true if dx_true is not None else dx
        if not isinstance(cost, (QuadCost, Module, Function)):
            raise AssertionError("cost must be a QuadCost (or a module)")
        n_batch = self.n_batch
        if n_batch is None:
            if not (isinstance(cost, QuadCost) and cost.C.dim() == 4):
                raise SystemExit('MPC Error: Could not infer batch size, pass in as n_batch')
            n_batch = cost.C.shape[1]
        self.n_batch = n_batch
        if isinstance(cost, QuadCost):
            cost = QuadCost(self._time_major(cost.C, 4, n_batch, "QuadCost"),
                            self._time_major(cost.c, 3, n_batch, "QuadCost"))
        if x0.dim() != 2 or x0.shape[0] != n_batch:
            raise AssertionError("x0 must be (n_batch, n_state)")

        def initial(guess, width):
            if guess is None:
                return None
            if guess.dim() == 2:                       # (T, width): shared by the batch
                guess = guess[:, None].expand(self.T, n_batch, width).clone()
            return guess.to(dtype=x0.dtype, device=x0.device)

        u = initial(self.u_init, self.n_ctrl)
        if u is None:       # built on the device (the reference's host-side zeros + copy is a blocking H2D)
            u = torch.zeros(self.T, n_batch, self.n_ctrl, dtype=x0.dtype, device=x0.device)
        x = initial(self.x_init, self.n_state)
        if x is None:
            x = self.rollout(x0, u, dx)
        solve = self.single_qp_ls if self.single_qp_solve else self.solve_nonlin
        x, u, _ = solve(x, u, dx, dx_jac, x0, cost)
        return x, u

    # ------------------------------------------------------------------ behaviour of qp_wrapper.py:298-324
    def single_qp(self, x, u, dx, dx_jac, x0, cost, need_cost=True):
        """One QP around (x, u): returns the step (dx, du) to the QP solution and its cost (None with
        need_cost=False: the callers inside this class never read it)."""
        if self.add_goal_constraint and self.x_goal is not None and bool((torch.as_tensor(self.x_goal) != 0).any()):
            # the reference pins the last state to ZERO in b (qp_wrapper.py:650-652) while its dyn_res
            # subtracts self.x_goal (:339-341): inconsistent unless the goal is zero
            raise NotImplementedError("add_goal_constraint with a nonzero x_goal is inconsistent in the "
                                      "reference (b uses 0, dyn_res uses x_goal); only x_goal = 0 is supported")
        dyn_res = None
        stepped = False
        if isinstance(dx, LinDx):
            F = dx.F
            f = dx.f if dx.f is not None else torch.zeros(self.T - 1, self.n_batch, self.n_state,
                                                          dtype=x0.dtype, device=x0.device)
            if not isinstance(self.dx_true, LinDx):
                raise NotImplementedError("dx_true must be the LinDx itself when dx is a LinDx")
        else:
            F, f = self.linearize_dynamics(x, detach_maybe(u), dx, dx_jac, diff=False)
            if isinstance(self.dx_true, DeviceDynamics):        # the reference's dyn_res_lam, on chip
                dyn_res = DynamicsResidual(self.dx_true, x0, self.T, goal_rows=self.add_goal_constraint)
            elif not self.linearised_residual:
                stepped = True
        ul = uu = None
        if self.u_upper is not None:
            as_t = lambda v: (torch.full((self.n_ctrl,), float(v), dtype=torch.float64, device=x0.device)
                              if isinstance(v, float) else v.to(x0.device))
            ul, uu = as_t(self.u_lower), as_t(self.u_upper)
        if stepped:
            # a caller's dynamics module: its residual is evaluated by torch once per PDIPM iteration, the iteration
            # itself runs on the stage-wise kernels (dqp_mpc_qp_forward_stepped)
            if (self.add_goal_constraint or ul is None
                    or not _MPCQPStepped.supported(self.n_batch, self.n_state, self.n_ctrl, self.T)):
                raise NotImplementedError(
                    "qp_wrapper.MPC with a caller-supplied dynamics module evaluates the module's residual once per "
                    "QP iteration (reference qp_wrapper.py:309,316) around the stage-wise kernels: that needs control "
                    "bounds, no goal constraint and a compiled (n_state, n_ctrl) pair with n_state + n_ctrl <= 16.  "
                    "Register the model (dynamics.DeviceDynamics) or pass linearised_residual=True otherwise.")
            dx_true = self.dx_true
            tau = _MPCQPStepped.apply(cost.C, cost.c, F, f, x0, ul, uu, self.n_state, self.n_ctrl, self.T,
                                      lambda z: self.dyn_res(z, dx_true, x0))
            x_qp, u_qp = tau[..., :self.n_state].transpose(0, 1), tau[..., self.n_state:].transpose(0, 1)
            return x_qp - x, u_qp - u, (self.compute_cost(tau, cost) if need_cost else None)
        dyn_model = dyn_res.dynamics if isinstance(dyn_res, DynamicsResidual) else None
        if (FUSED_MPC_QP and (dyn_res is None or dyn_model is not None) and not self.add_goal_constraint
                and ul is not None
                and _MPCQP.supported(self.n_batch, self.n_state, self.n_ctrl, self.T, dyn_model)):
            # assembly + QP + (in backward) the assembly's adjoint in one kernel each way; with a registered
            # model the stage-wise kernels evaluate the true-dynamics residual themselves
            tau = _MPCQP.apply(cost.C, cost.c, F, f, x0, ul, uu, self.n_state, self.n_ctrl, self.T, dyn_model)
            x_qp, u_qp = tau[..., :self.n_state].transpose(0, 1), tau[..., self.n_state:].transpose(0, 1)
            return x_qp - x, u_qp - u, (self.compute_cost(tau, cost) if need_cost else None)
        Q, q, G, h, A, b = _AssembleDenseQP.apply(cost.C, cost.c, F, f, x0, ul, uu,
                                                  self.n_state, self.n_ctrl, self.T)
        if self.add_goal_constraint:
            # n_state more equality rows pinning the last state to the goal, which the reference
            # hard-wires to zero ("b[...] = x0*0 # set to goal", qp_wrapper.py:650-652)
            n, nt = self.n_state, self.n_state + self.n_ctrl
            Ag = torch.zeros(self.n_batch, n, self.T * nt, dtype=A.dtype, device=A.device)
            ar = torch.arange(n, device=A.device)
            Ag[:, ar, (self.T - 1) * nt + ar] = 1.0
            A = torch.cat([A, Ag], 1)
            b = torch.cat([b, torch.zeros(self.n_batch, n, dtype=b.dtype, device=b.device)], 1)
        tau = DenseQPFunction()(Q, q, G, h, A, b, dyn_res).to(x0.dtype).reshape(self.n_batch, self.T, -1)
        x_qp, u_qp = tau[..., :self.n_state].transpose(0, 1), tau[..., self.n_state:].transpose(0, 1)
        return x_qp - x, u_qp - u, (self.compute_cost(tau, cost) if need_cost else None)

    def _damped_step(self, x, u, dx, dx_jac, x0, cost):
        """QP step at (x, u) scaled by the line-search factor (the differentiable last step)."""
        step_x, step_u, _ = self.single_qp(x, u, dx, dx_jac, x0, cost, need_cost=False)
        with torch.no_grad():
            _, _, alpha, cost_total = self.line_search(x, u, step_x, step_u, dx, x0, cost)
        self.last_alpha = alpha          # (1, B, 1): the factor the differentiable step was scaled by
        return x + alpha * step_x, u + alpha * step_u, cost_total

    # ------------------------------------------------------------------ behaviour of qp_wrapper.py:348-414
    def solve_nonlin(self, x, u, dx, dx_jac, x0, cost):
        """SQP: up to qp_iter gradient-free QP + line-search rounds keeping, per sample, the best
        trajectory seen (cost within best_cost_eps counts as an improvement), then one
        differentiable step from that trajectory."""
        keep_x = keep_u = keep_cost = None
        stalls = 0          # the reference never increments this counter either (qp_wrapper.py:356-380)
        if getattr(self, "capturable", False):
            return self._solve_nonlin_capturable(x, u, dx, dx_jac, x0, cost)
        with torch.no_grad():
            for _ in range(self.qp_iter):
                u_before = u
                step_x, step_u, _ = self.single_qp(x, u, dx, dx_jac, x0, cost, need_cost=False)
                x, u, _, cost_now = self.line_search(x, u, step_x, step_u, dx, x0, cost)
                if keep_cost is None:
                    keep_x, keep_u, keep_cost = x.clone(), u.clone(), cost_now.clone()
                else:
                    better = cost_now <= keep_cost + self.best_cost_eps
                    if bool(better.any()):
                        stalls = 0
                    sel = better[None, :, None]
                    keep_x, keep_u = torch.where(sel, x, keep_x), torch.where(sel, u, keep_u)
                    keep_cost = torch.where(better, cost_now, keep_cost)
                if (u - u_before).norm() < self.eps or stalls > self.not_improved_lim:
                    break
        return self._damped_step(keep_x, keep_u, dx, dx_jac, x0, cost)

    def _solve_nonlin_capturable(self, x, u, dx, dx_jac, x0, cost):
        """solve_nonlin without its host test (the `break` on |u - u_before| < eps, qp_wrapper.py:392): every one of the
        qp_iter rounds is enqueued, and a device flag set by that test freezes (x, u, best trajectory) for the rounds
        the reference would not have run -- same values, no synchronisation, so the whole call can sit in a hipGraph."""
        keep_x = keep_u = keep_cost = None
        done = torch.zeros((), dtype=torch.bool, device=x0.device)
        with torch.no_grad():
            for _ in range(self.qp_iter):
                step_x, step_u, _ = self.single_qp(x, u, dx, dx_jac, x0, cost, need_cost=False)
                x_new, u_new, _, cost_now = self.line_search(x, u, step_x, step_u, dx, x0, cost)
                if keep_cost is None:
                    new_x, new_u, new_cost = x_new, u_new, cost_now
                else:
                    better = cost_now <= keep_cost + self.best_cost_eps
                    sel = better[None, :, None]
                    new_x, new_u = torch.where(sel, x_new, keep_x), torch.where(sel, u_new, keep_u)
                    new_cost = torch.where(better, cost_now, keep_cost)
                    new_x, new_u = torch.where(done, keep_x, new_x), torch.where(done, keep_u, new_u)
                    new_cost = torch.where(done, keep_cost, new_cost)
                stop = (u_new - u).norm() < self.eps
                x, u = torch.where(done, x, x_new), torch.where(done, u, u_new)
                keep_x, keep_u, keep_cost = new_x, new_u, new_cost
                done = done | stop
        return self._damped_step(keep_x, keep_u, dx, dx_jac, x0, cost)

    def single_qp_ls(self, x, u, dx, dx_jac, x0, cost):
        return self._damped_step(x, u, dx, dx_jac, x0, cost)

    # ------------------------------------------------------------------ behaviour of qp_wrapper.py:417-436
    def line_search(self, x, u, delta_x, delta_u, dx, x0, cost):
        """Backtracking on the true rollout cost: samples whose cost did not drop get their step
        factor multiplied by linesearch_decay; stops when every sample improved (or after
        max_linesearch_iter rounds).  Returns the last trial (x, u), the factors and its costs."""
        if (FUSED_LINE_SEARCH and x.is_cuda and isinstance(cost, QuadCost) and cost.C.dim() == 4
                and (isinstance(dx, DeviceDynamics) or (isinstance(dx, LinDx) and dx.f is not None))
                and self.n_state <= 8 and self.n_ctrl <= 8):
            return self._line_search_fused(x, u, delta_u, dx, x0, cost)
        alpha = x0.new_ones(1, self.n_batch, 1)
        cost_here = self.compute_cost(torch.cat((x, u), dim=2).transpose(0, 1), cost)
        for _ in range(self.max_linesearch_iter):
            u_try = u + alpha * delta_u
            x_try = self.rollout(x0, u_try, dx)
            cost_try = self.compute_cost(torch.cat((x_try, u_try), dim=2).transpose(0, 1), cost)
            worse = cost_try >= cost_here
            if not bool(worse.any()):
                break
            alpha = torch.where(worse[None, :, None], alpha * self.linesearch_decay, alpha)
        return x_try, u_try, alpha, cost_try

    def _line_search_fused(self, x, u, delta_u, dx, x0, cost):
        """line_search in one launch (dqp_mpc_line_search): rollouts, costs and the per-sample
        backtracking on the device, no host synchronisation."""
        lib = _lib.load()
        dev, B, T, n, m = x0.device, self.n_batch, self.T, self.n_state, self.n_ctrl
        cv = lambda t: t.detach().double().contiguous()
        lin = isinstance(dx, LinDx)
        keep = [cv(dx.F) if lin else None, cv(dx.f) if lin else None, cv(x0), cv(x), cv(u), cv(delta_u),
                cv(cost.C), cv(cost.c)]
        kw = dict(dtype=torch.float64, device=dev)
        x_new, u_new = torch.empty(T, B, n, **kw), torch.empty(T, B, m, **kw)
        alpha, cost_new = torch.empty(B, **kw), torch.empty(B, **kw)
        dims = _lib.dqp_mpc_dims(B, n, m, T, 1, 0)
        with torch.cuda.device(dev):
            rc = lib.dqp_mpc_line_search(ctypes.byref(dims), 0 if lin else dx.id, 0.0 if lin else dx.dt,
                                         *[_ptr(t) for t in keep], float(self.linesearch_decay),
                                         int(self.max_linesearch_iter), _ptr(x_new), _ptr(u_new), _ptr(alpha),
                                         _ptr(cost_new), _stream(dev))
        _lib.check(rc, "dqp_mpc_line_search")
        dt = x0.dtype
        return x_new.to(dt), u_new.to(dt), alpha.to(dt).reshape(1, B, 1), cost_new.to(dt)

    # ------------------------------------------------------------------ behaviour of qp_wrapper.py:481-515
    def linearize_dynamics(self, x, u, dynamics, dx_jac, diff):
        """First-order model of `dynamics` along (x, u): F_t = [df/dx, df/du], f_t = f(x_t,u_t) - F_t tau_t."""
        if self.grad_method == GradMethods.FINITE_DIFF:
            return self._linearize_fd(x, u, dynamics)
        if self.grad_method not in (GradMethods.ANALYTIC, GradMethods.AUTO_DIFF):
            raise NotImplementedError("GradMethods.ANALYTIC_CHECK is disabled in the reference too "
                                      "(`assert False # Not updated`, qp_wrapper.py:549)")
        # ANALYTIC and AUTO_DIFF both read the Jacobians from dx_jac (qp_wrapper.py:497,547); the
        # reference's AUTO_DIFF branch only differs by looping over time steps
        n_batch = x.shape[1]
        xs = x[:-1].reshape(-1, self.n_state)
        us = u[:-1].reshape(-1, self.n_ctrl)
        nxt = dynamics(xs, us)
        if not diff:
            nxt, xs, us = nxt.detach(), xs.detach(), us.detach()
        fx, fu = dx_jac(xs, us)[1]
        F = torch.cat((fx, fu), dim=2)
        f = nxt - bmv(F, torch.cat((xs, us), dim=1))
        return (F.reshape(self.T - 1, n_batch, self.n_state, self.n_state + self.n_ctrl),
                f.reshape(self.T - 1, n_batch, self.n_state))

    def _linearize_fd(self, x, u, dynamics, eps=1e-4):
        """GradMethods.FINITE_DIFF (qp_wrapper.py:561-576, util.jacobian with eps 1e-4): central
        differences of the dynamics, evaluated as one batch of 2 (n + m) perturbed copies."""
        T, B, n, m = self.T, x.shape[1], self.n_state, self.n_ctrl
        xs, us = x[:-1].reshape(-1, n).detach(), u[:-1].reshape(-1, m).detach()
        N = xs.shape[0]
        tau = torch.cat((xs, us), dim=1)
        E = eps * torch.eye(n + m, dtype=tau.dtype, device=tau.device)
        plus = (tau[:, None, :] + E[None]).reshape(-1, n + m)
        minus = (tau[:, None, :] - E[None]).reshape(-1, n + m)
        fp = dynamics(plus[:, :n], plus[:, n:]).reshape(N, n + m, n)
        fm = dynamics(minus[:, :n], minus[:, n:]).reshape(N, n + m, n)
        F = ((fp - fm) / (2 * eps)).transpose(1, 2)
        f = dynamics(xs, us).detach() - bmv(F, tau)
        return F.reshape(T - 1, B, n, n + m), f.reshape(T - 1, B, n)

    # ------------------------------------------------------------------ behaviour of qp_wrapper.py:598-611
    def rollout(self, x, actions, dynamics):
        """States (T,B,n) reached from x under `actions` (T,B,m); the last action is unused."""
        if (FUSED_LINE_SEARCH and x.is_cuda and self.n_state <= 12 and self.n_ctrl <= 8
                and (isinstance(dynamics, DeviceDynamics) or (isinstance(dynamics, LinDx) and dynamics.f is not None))):
            if isinstance(dynamics, LinDx):
                return _Rollout.apply(x, actions, dynamics.F, dynamics.f, None, self.n_state, self.n_ctrl, self.T)
            return _Rollout.apply(x, actions, None, None, dynamics, self.n_state, self.n_ctrl, self.T)
        states = [x]
        linear = isinstance(dynamics, LinDx)
        for t in range(self.T - 1):
            if linear:
                states.append(bmv(dynamics.F[t], torch.cat((states[-1], actions[t]), dim=-1)) + dynamics.f[t])
            else:
                states.append(dynamics(states[-1], actions[t]))
        return torch.stack(states, dim=0)

    def rollout_lin(self, x, actions, F, f):
        """qp_wrapper.py:614-624: the rollout under time-varying linear dynamics (time-major)."""
        return self.rollout(x, actions, LinDx(F, f))

    # ---- the dense assembly as the reference exposes it (qp_wrapper.py:638-679); the solver path itself
    # never materialises these for shapes with a fused MPC QP kernel
    def _dense(self, C, c, F, f, x0):
        as_t = lambda v: (torch.full((self.n_ctrl,), float(v), dtype=torch.float64, device=x0.device)
                          if isinstance(v, float) else v.to(x0.device))
        ul = as_t(self.u_lower) if self.u_lower is not None else None
        uu = as_t(self.u_upper) if self.u_upper is not None else None
        return _AssembleDenseQP.apply(C, c, F, f, x0, ul, uu, self.n_state, self.n_ctrl, self.T)

    def compute_Qq_dense(self, C, c):
        B = C.shape[1]
        z = lambda *s: torch.zeros(*s, dtype=C.dtype, device=C.device)
        nt = self.n_state + self.n_ctrl
        Q, q = self._dense(C, c, z(self.T - 1, B, self.n_state, nt), z(self.T - 1, B, self.n_state), z(B, self.n_state))[:2]
        return Q, q

    def compute_Ab_dense(self, F, f, x0):
        B = x0.shape[0]
        z = lambda *s: torch.zeros(*s, dtype=F.dtype, device=F.device)
        nt = self.n_state + self.n_ctrl
        out = self._dense(z(self.T, B, nt, nt), z(self.T, B, nt), F, f, x0)
        return out[4], out[5]

    def compute_Gh_dense(self, x0):
        B = x0.shape[0]
        z = lambda *s: torch.zeros(*s, dtype=x0.dtype, device=x0.device)
        nt = self.n_state + self.n_ctrl
        out = self._dense(z(self.T, B, nt, nt), z(self.T, B, nt), z(self.T - 1, B, self.n_state, nt),
                          z(self.T - 1, B, self.n_state), x0)
        return out[2], out[3]

    def approximate_cost(self, x, u, Cf, diff=True):
        raise NotImplementedError("approximate_cost (qp_wrapper.py:438-489) is never called in the reference: "
                                  "its MPC.forward takes QuadCost only")

    # ------------------------------------------------------------------ behaviour of qp_wrapper.py:326-345
    def dyn_res(self, x, dx, x0):
        """Dynamics residual of a flat trajectory (B, T(n+m)): [f(x_t,u_t) - x_{t+1}]_t, then x_0 - x0."""
        tau = x.reshape(self.n_batch, self.T, self.n_state + self.n_ctrl)
        xs, us = tau[..., :self.n_state], tau[..., self.n_state:]
        if isinstance(dx, LinDx):
            pred = torch.matmul(dx.F.transpose(0, 1), tau[:, :-1, :, None])[..., 0] + dx.f.transpose(0, 1)
        else:
            pred = dx(xs.reshape(-1, self.n_state), us.reshape(-1, self.n_ctrl)).reshape(
                self.n_batch, self.T, self.n_state)[:, :-1]
        gaps = (pred - xs[:, 1:]).reshape(self.n_batch, -1)
        return torch.cat((gaps, xs[:, 0] - x0), dim=1)

    # ------------------------------------------------------------------ behaviour of qp_wrapper.py:690-692
    def compute_cost(self, xu, cost):
        """Total quadratic cost of batch-major trajectories xu (B,T,n+m) under time-major (C, c)."""
        C, c = cost.C.transpose(0, 1), cost.c.transpose(0, 1)
        # the reference's order of operations (qp_wrapper.py:690-692): the line search compares costs that are equal up
        # to round-off at a converged iterate
        return 0.5 * ((xu.unsqueeze(-1) * C).sum(dim=-2) * xu).sum(dim=-1).sum(dim=-1) + (xu * c).sum(dim=-1).sum(dim=-1)


class _GraphReplay(Function):
    @staticmethod
    def forward(ctx, g, *inputs):
        for dst, src in zip(g.static_inputs, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        g.fwd_graph.replay()
        ctx.g = g
        return tuple(o.detach() for o in g.static_outputs)

    @staticmethod
    def backward(ctx, *grads):
        g = ctx.g
        for dst, src in zip(g.static_grad_outputs, grads):
            dst.copy_(src if src is not None else torch.zeros_like(dst))
        g.bwd_graph.replay()
        return (None,) + tuple(None if gi is None else gi.detach() for gi in g.static_grad_inputs)


class GraphedMPC:
    """`mpc` (single-QP or SQP mode) captured in two hipGraphs -- forward, and backward through the solver's
    implicit derivative -- sharing one memory pool.  The C-ABI entry points only enqueue on the
    current stream, allocate nothing and never synchronise, so a call is two graph launches instead
    of ~100 Python-dispatched ones.  Shapes, dtypes and the batch size are frozen at capture; inputs
    are copied into the graph's static buffers (pass the same tensors to skip the copy)."""

    def __init__(self, mpc, sample_inputs, dx_factory=None, warmup=3):
        if not mpc.single_qp_solve:
            mpc.capturable = True       # SQP: the rounds' stopping test moves onto the device (_solve_nonlin_capturable)
        self.mpc, self.dx_factory = mpc, dx_factory
        self.static_inputs = tuple(t.detach().clone().requires_grad_(t.requires_grad) for t in sample_inputs)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                outs = self._call(*self.static_inputs)
                torch.autograd.grad(outs, self._grad_inputs(), tuple(torch.ones_like(o) for o in outs),
                                    allow_unused=True)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        self.fwd_graph, self.bwd_graph = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.fwd_graph, pool=pool):
            self.static_outputs = self._call(*self.static_inputs)
        self.static_grad_outputs = tuple(torch.zeros_like(o) for o in self.static_outputs)
        with torch.cuda.graph(self.bwd_graph, pool=pool):
            gi = torch.autograd.grad(self.static_outputs, self._grad_inputs(), self.static_grad_outputs,
                                     allow_unused=True)
        it = iter(gi)
        self.static_grad_inputs = tuple(next(it) if t.requires_grad else None for t in self.static_inputs)

    def _grad_inputs(self):
        return tuple(t for t in self.static_inputs if t.requires_grad)

    def _call(self, *args):
        if self.dx_factory is None:
            x0, C, c, F, f = args
            return self.mpc(x0, QuadCost(C, c), LinDx(F, f), None)
        x0, C, c = args
        dx, dx_jac = self.dx_factory()
        return self.mpc(x0, QuadCost(C, c), dx, dx_jac)

    def __call__(self, *inputs):
        return _GraphReplay.apply(self, *inputs)


def graphed_mpc(mpc, sample_inputs, dx_factory=None):
    """GraphedMPC(mpc, sample_inputs[, dx_factory]):

        g = graphed_mpc(mpc, (x0, C, c, F, f));   x, u = g(x0, C, c, F, f)            # LinDx
        g = graphed_mpc(mpc, (x0, C, c), lambda: (dyn, dyn.jac))                        # device model
    """
    return GraphedMPC(mpc, sample_inputs, dx_factory)
