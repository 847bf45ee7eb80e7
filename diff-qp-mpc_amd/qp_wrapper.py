"""qp_wrapper.MPC with the reference's interface (qpth/qp_wrapper.py:22-34, 59-692).

    MPC(n_state, n_ctrl, T, u_lower=..., u_upper=..., qp_iter=..., single_qp_solve=..., ...)
        (x0, QuadCost(C, c), LinDx(F, f) | dynamics callable, dx_jac[, dx_true]) -> (x, u)

time-major tensors, as in the reference: C (T,B,nt,nt) c (T,B,nt) F (T-1,B,n,nt) f (T-1,B,n)
x0 (B,n); returns x (T,B,n), u (T,B,m).

What runs where:
  * dense QP assembly (compute_Qq/Ab/Gh_dense, qp_wrapper.py:638-679) and its adjoint: HIP
    kernels behind dqp_mpc_assemble / dqp_mpc_assemble_backward (csrc/dqp_mpc.hip);
  * the QP itself (qp.DenseQPFunction, qp_wrapper.py:316): fused HIP PDIPM (csrc/dqp_*.hip);
  * the SQP outer loop, rollout, cost and line search (qp_wrapper.py:298-436, 598-611, 690-692):
    torch ops on the same device, statement-for-statement the reference's control flow
    (they call user-supplied Python dynamics, which cannot be fused).

Difference from the reference: it passes the TRUE-dynamics residual closure into the PDIPM
(`dyn_res_lam`, qp_wrapper.py:309,316); the fused kernel evaluates the residual of the
LINEARISED dynamics, A z - b.  For LinDx the two are identical (this is what the parity tests
pin); for nonlinear dynamics they differ inside the QP iterations only.
"""
import ctypes
import sys
from collections import namedtuple
from enum import Enum

import torch
from torch.autograd import Function
from torch.nn import Module

from . import _lib
from .qp import DenseQPFunction

QuadCost = namedtuple('QuadCost', 'C c')
LinDx = namedtuple('LinDx', 'F f')
QuadCost.__new__.__defaults__ = (None,) * len(QuadCost._fields)
LinDx.__new__.__defaults__ = (None,) * len(LinDx._fields)


class GradMethods(Enum):
    AUTO_DIFF = 1
    FINITE_DIFF = 2
    ANALYTIC = 3
    ANALYTIC_CHECK = 4


def detach_maybe(x):
    """qpth/util.py:204-207"""
    if x is None:
        return None
    return x if not x.requires_grad else x.detach()


def bmv(X, y):
    """qpth/util.py:92-93"""
    return X.bmm(y.unsqueeze(2)).squeeze(2)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


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
                 single_qp_solve=False, add_goal_constraint=False, x_goal=None):
        super().__init__()
        assert (u_lower is None) == (u_upper is None)
        assert max_linesearch_iter > 0
        if solver_type != 'dense':
            raise NotImplementedError("only solver_type='dense' exists in the reference too")
        self.n_state, self.n_ctrl, self.T = n_state, n_ctrl, T
        self.u_lower, self.u_upper, self.x_goal = u_lower, u_upper, x_goal
        if not isinstance(u_lower, float):
            self.u_lower = detach_maybe(self.u_lower)
        if not isinstance(u_upper, float):
            self.u_upper = detach_maybe(self.u_upper)
        self.u_zero_I = detach_maybe(u_zero_I)
        self.u_init = detach_maybe(u_init)
        self.x_init = detach_maybe(x_init)
        self.qp_iter = qp_iter
        self.grad_method = grad_method
        self.delta_u = delta_u
        self.verbose = verbose
        self.eps = eps
        self.back_eps = back_eps
        self.n_batch = n_batch
        self.linesearch_decay = linesearch_decay
        self.max_linesearch_iter = max_linesearch_iter
        self.exit_unconverged = exit_unconverged
        self.detach_unconverged = detach_unconverged
        self.backprop = backprop
        self.not_improved_lim = not_improved_lim
        self.best_cost_eps = best_cost_eps
        self.slew_rate_penalty = slew_rate_penalty
        self.prev_ctrl = prev_ctrl
        self.solver_type = solver_type
        self.single_qp_solve = single_qp_solve
        self.add_goal_constraint = add_goal_constraint

    # ------------------------------------------------------------------ qp_wrapper.py:213-296
    def forward(self, x0, cost, dx, dx_jac, dx_true=None):
        self.dx_true = dx if dx_true is None else dx_true
        assert isinstance(cost, QuadCost) or isinstance(cost, Module) or isinstance(cost, Function)
        if self.n_batch is not None:
            n_batch = self.n_batch
        elif isinstance(cost, QuadCost) and cost.C.ndimension() == 4:
            n_batch = cost.C.size(1)
        else:
            print('MPC Error: Could not infer batch size, pass in as n_batch')
            sys.exit(-1)
        self.n_batch = n_batch      # the reference reads self.n_batch in single_qp/dyn_res

        if isinstance(cost, QuadCost):
            C, c = cost
            if C.ndimension() == 2:
                C = C.unsqueeze(0).unsqueeze(0).expand(self.T, n_batch, self.n_state + self.n_ctrl, -1)
            elif C.ndimension() == 3:
                C = C.unsqueeze(1).expand(self.T, n_batch, self.n_state + self.n_ctrl, -1)
            if c.ndimension() == 1:
                c = c.unsqueeze(0).unsqueeze(0).expand(self.T, n_batch, -1)
            elif c.ndimension() == 2:
                c = c.unsqueeze(1).expand(self.T, n_batch, -1)
            if C.ndimension() != 4 or c.ndimension() != 3:
                print('MPC Error: Unexpected QuadCost shape.')
                sys.exit(-1)
            cost = QuadCost(C, c)

        assert x0.ndimension() == 2 and x0.size(0) == n_batch
        if self.u_init is None:
            u = torch.zeros(self.T, n_batch, self.n_ctrl, dtype=x0.dtype, device=x0.device)   # on device:
            # the reference builds it on the host and copies (qp_wrapper.py:174), a blocking H2D per call
        else:
            u = self.u_init
            if u.ndimension() == 2:
                u = u.unsqueeze(1).expand(self.T, n_batch, -1).clone()
        u = u.type_as(x0.data)
        if self.x_init is None:
            x = self.rollout(x0, u, dx)
        else:
            x = self.x_init
            if x.ndimension() == 2:
                x = x.unsqueeze(1).expand(self.T, n_batch, -1).clone()
        x = x.type_as(x0.data)

        if self.single_qp_solve:
            x, u, cost_total = self.single_qp_ls(x, u, dx, dx_jac, x0, cost)
        else:
            x, u, cost_total = self.solve_nonlin(x, u, dx, dx_jac, x0, cost)
        return (x, u)

    # ------------------------------------------------------------------ qp_wrapper.py:298-324
    def single_qp(self, x, u, dx, dx_jac, x0, cost):
        if isinstance(dx, LinDx):
            F, f = dx.F, dx.f
            if f is None:
                f = torch.zeros((self.T - 1, self.n_batch, self.n_state), dtype=x0.dtype, device=x0.device)
        else:
            F, f = self.linearize_dynamics(x, detach_maybe(u), dx, dx_jac, diff=False)
        bounds = self.u_upper is not None
        ul = uu = None
        if bounds:
            as_t = lambda v: (torch.full((self.n_ctrl,), float(v), dtype=torch.float64, device=x0.device)
                              if isinstance(v, float) else v.to(x0.device))
            ul, uu = as_t(self.u_lower), as_t(self.u_upper)
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
        xhats_qpf = DenseQPFunction()(Q, q, G, h, A, b, None).to(x0.dtype)
        xhats_qpf = xhats_qpf.reshape(self.n_batch, self.T, -1)
        x_hat = xhats_qpf[:, :, :self.n_state].transpose(0, 1)
        u_hat = xhats_qpf[:, :, self.n_state:].transpose(0, 1)
        cost_total = self.compute_cost(xhats_qpf, cost)
        return x_hat - x, u_hat - u, cost_total

    # ------------------------------------------------------------------ qp_wrapper.py:348-414
    def solve_nonlin(self, x, u, dx, dx_jac, x0, cost):
        best = None
        n_not_improved = 0
        with torch.no_grad():
            for i in range(self.qp_iter):
                u_prev = u.clone()
                delta_x, delta_u, _ = self.single_qp(x, u, dx, dx_jac, x0, cost)
                x, u, alpha, cost_total = self.line_search(x, u, delta_x, delta_u, dx, x0, cost)
                full_du_norm = (u - u_prev).norm()
                if best is None:
                    best = {'x': x.clone(), 'u': u.clone(), 'costs': cost_total.clone()}
                else:
                    # per-sample best (the reference loops over the batch in Python,
                    # qp_wrapper.py:372-377; same selection, vectorised)
                    I = cost_total <= best['costs'] + self.best_cost_eps
                    if bool(I.any()):
                        n_not_improved = 0
                    best['x'][:, I] = x[:, I]
                    best['u'][:, I] = u[:, I]
                    best['costs'][I] = cost_total[I]
                if full_du_norm < self.eps or n_not_improved > self.not_improved_lim:
                    break
        x, u = best['x'], best['u']
        delta_x, delta_u, _ = self.single_qp(x, u, dx, dx_jac, x0, cost)
        with torch.no_grad():
            _, _, alpha, cost_total = self.line_search(x, u, delta_x, delta_u, dx, x0, cost)
        x = x + delta_x * alpha
        u = u + delta_u * alpha
        return x, u, cost_total

    def single_qp_ls(self, x, u, dx, dx_jac, x0, cost):
        delta_x, delta_u, _ = self.single_qp(x, u, dx, dx_jac, x0, cost)
        with torch.no_grad():
            _, _, alpha, cost_total = self.line_search(x, u, delta_x, delta_u, dx, x0, cost)
        x = x + delta_x * alpha
        u = u + delta_u * alpha
        return x, u, cost_total

    # ------------------------------------------------------------------ qp_wrapper.py:417-436
    def line_search(self, x, u, delta_x, delta_u, dx, x0, cost):
        alpha = torch.ones([1, self.n_batch, 1], dtype=x0.dtype, device=x0.device)
        cost_total = self.compute_cost(torch.cat((x, u), dim=2).transpose(0, 1), cost)
        for j in range(self.max_linesearch_iter):
            u_new = u + delta_u * alpha
            x_new = self.rollout(x0, u_new, dx)
            xhats_qpf = torch.cat((x_new, u_new), dim=2).transpose(0, 1)
            cost_total_new = self.compute_cost(xhats_qpf, cost)
            if (cost_total_new < cost_total).all():
                break
            else:
                mask = (cost_total_new >= cost_total).to(alpha.dtype)[None, :, None]
                alpha = alpha * self.linesearch_decay * mask + (1 - mask) * alpha
        return x_new, u_new, alpha, cost_total_new

    # ------------------------------------------------------------------ qp_wrapper.py:481-515
    def linearize_dynamics(self, x, u, dynamics, dx_jac, diff):
        if self.grad_method != GradMethods.ANALYTIC:
            raise NotImplementedError("only GradMethods.ANALYTIC (dx_jac) is mirrored")
        n_batch = x[0].size(0)
        _u = u[:-1].reshape(-1, self.n_ctrl)
        _x = x[:-1].contiguous().view(-1, self.n_state)
        _new_x = dynamics(_x, _u)
        if not diff:
            _new_x, _x, _u = _new_x.detach(), _x.detach(), _u.detach()
        R, S = dx_jac(_x, _u)[1]
        f = _new_x - bmv(R, _x) - bmv(S, _u)
        f = f.view(self.T - 1, n_batch, self.n_state)
        R = R.contiguous().view(self.T - 1, n_batch, self.n_state, self.n_state)
        S = S.contiguous().view(self.T - 1, n_batch, self.n_state, self.n_ctrl)
        F = torch.cat((R, S), 3)
        return F, f

    # ------------------------------------------------------------------ qp_wrapper.py:598-611
    def rollout(self, x, actions, dynamics):
        x = [x]
        for t in range(self.T - 1):
            xt, ut = x[t], actions[t]
            if isinstance(dynamics, LinDx):
                new_x = bmv(dynamics.F[t], torch.cat([xt, ut], dim=-1)) + dynamics.f[t]
            else:
                new_x = dynamics(xt, ut)
            x.append(new_x)
        return torch.stack(x, 0)

    # ------------------------------------------------------------------ qp_wrapper.py:326-345
    def dyn_res(self, x, dx, x0):
        x = x.reshape(self.n_batch, self.T, self.n_state + self.n_ctrl)
        x, u = x[:, :, :self.n_state], x[:, :, self.n_state:]
        if isinstance(dx, LinDx):
            x_next = (dx.F.permute(1, 0, 2, 3) * torch.cat((x, u), dim=2)[:, :-1, None, :]).sum(dim=-1) \
                + dx.f.permute(1, 0, 2)
        else:
            x_next = dx(x.reshape(-1, self.n_state), u.reshape(-1, self.n_ctrl)).reshape(
                self.n_batch, self.T, self.n_state)[:, :-1]
        res = (x_next - x[:, 1:, :]).reshape(self.n_batch, -1)
        res_init = (x[:, 0, :] - x0).reshape(self.n_batch, -1)
        return torch.cat((res, res_init), dim=1)

    # ------------------------------------------------------------------ qp_wrapper.py:690-692
    def compute_cost(self, xu, cost):
        C = cost.C.transpose(0, 1)
        c = cost.c.transpose(0, 1)
        return 0.5 * ((xu.unsqueeze(-1) * C).sum(dim=-2) * xu).sum(dim=-1).sum(dim=-1) + \
            (xu * c).sum(dim=-1).sum(dim=-1)
