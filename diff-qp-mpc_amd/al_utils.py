"""Augmented-Lagrangian Newton solver, mirror of qpth/al_utils.py.

Public names follow the reference so call sites read the same:
    QuadCost, LinDx                      al_utils.py:8-13
    NewtonAL.apply(meritfn, dyn_fn, cost_fn, merit_grad_hessfn, xi, x0, lam, rho, Q, q,
                   threshold, eps, ls) -> (x_est, status)           al_utils.py:363-500
    merit_function, merit_grad_hessian, dyn_res, compute_cost, compute_cost_gradient,
    line_search_newton, warm_start_al                               al_utils.py:16-360,503-527

What runs where.  The Hessian assembly diag(Q) + rho Jc^T Jc, its Cholesky factorisation and the
Newton solve (al_utils.py:96-102,414-418) and the backward solve (al_utils.py:477-480) are the
HIP kernels of csrc/dqp_al.hip behind dqp_al_newton_step / dqp_al_chol_solve.  Residuals,
Jacobian blocks, the merit function and the 20-way line search call the user's Python dynamics
(`dx`, `dx_jac`), exactly as the reference does, and are device-tensor torch ops.

Differences from the reference: `merit_grad_hessian` returns (grad, HessianTerms) -- the pieces
the kernel consumes (clamped Jacobian, cost diagonal, rho) -- instead of a materialised
(B,nz,nz) Hessian; `HessianTerms.dense()` builds it when the rare LU fallback needs it.
Only diagonal costs (`diag_cost=True`, the only mode AL_mpc.MPC.forward reaches, AL_mpc.py:247-248)
are supported.
"""
import ctypes
from collections import namedtuple

import torch

from . import _lib

QuadCost = namedtuple("QuadCost", "C c")
LinDx = namedtuple("LinDx", "F f")
QuadCost.__new__.__defaults__ = (None,) * len(QuadCost._fields)
LinDx.__new__.__defaults__ = (None,) * len(LinDx._fields)

# NewtonALDevice: True = the block-tridiagonal Newton step (one launch, no Jacobian / Hessian in HBM),
# False = dense Jacobian + MFMA Hessian + dense Cholesky (the reference's own formulation)
BANDED_NEWTON_AL = True
N_LINESEARCH = 20          # al_utils.py:504
MAX_NEWTON_STEPS = 4       # al_utils.py:391


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _need_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("diff_qp_mpc_amd AL solver runs only on a GPU (HIP); there is no CPU fallback.")


# ------------------------------------------------------------------------------------------
# residuals and Jacobians of  x_{t+1} = f(x_t, u_t), x_0 = x0, u_lower <= u <= u_upper
# ------------------------------------------------------------------------------------------
def _split(xu, n_state):
    return xu[..., :n_state], xu[..., n_state:]


def _step(dx, x, u):
    """x_next for the first T-1 knots; dx is a callable or a LinDx (AL_mpc.py:336-339)."""
    B, T, n = x.shape
    m = u.shape[-1]
    if isinstance(dx, LinDx):
        xu = torch.cat((x, u), dim=2)[:, :-1]
        return (dx.F.permute(1, 0, 2, 3) * xu[:, :, None, :]).sum(-1) + dx.f.permute(1, 0, 2)
    return dx(x[:, :-1].reshape(-1, n), u[:, :-1].reshape(-1, m)).view(B, T - 1, n)


def dyn_res_eq(x, u, dx, x0):
    """[x_{t+1} - f(x_t,u_t)]_{t<T-1} then x_0 - x0   (al_utils.py:188-205)."""
    B = x.shape[0]
    gap = x[:, 1:] - _step(dx, x, u)
    return torch.cat((gap, (x[:, :1] - x0[:, None])), dim=1).reshape(B, -1)


def dyn_res_ineq(x, u, x0, x_lower, x_upper, u_lower, u_upper):
    """per knot [u - u_upper, u_lower - u]; clamp at 0 for the penalty term (al_utils.py:266-291)."""
    B = x.shape[0]
    res = torch.cat((u - u_upper, u_lower - u), dim=2).reshape(B, -1)
    return res, res.clamp(min=0)


def dyn_res(xu, dx, x0, x_lower=None, x_upper=None, u_lower=None, u_upper=None):
    """al_utils.py:321-336 -> (res, res_clamp), each (B, neq + nineq)."""
    x, u = _split(xu, x0.shape[-1])
    eq = dyn_res_eq(x, u, dx, x0)
    iq, iqc = dyn_res_ineq(x, u, x0, x_lower, x_upper, u_lower, u_upper)
    return torch.cat((eq, iq), 1), torch.cat((eq, iqc), 1)


def constraint_jacobian(xu, x0, dx_jac, u_lower, u_upper):
    """Residuals and the dense constraint Jacobian (al_utils.py:162-186,212-318).

    Returns res, res_clamp (B,ncon) and J, Jc (B,ncon,nz); Jc has the rows of inactive
    inequalities (res_clamp == 0) zeroed.  Row/column order as in the reference: equality rows
    knot-major then the x_0 block, inequality rows per knot [upper (m), lower (m)]; columns
    per knot [x_t, u_t]."""
    n = x0.shape[-1]
    B, T, nt = xu.shape
    m = nt - n
    x, u = _split(xu, n)
    x_next, (Jx, Ju) = dx_jac(x[:, :-1].reshape(-1, n), u[:, :-1].reshape(-1, m))
    Jx = Jx.reshape(B, T - 1, n, n)
    Ju = Ju.reshape(B, T - 1, n, m)
    eq = torch.cat((x[:, 1:] - x_next.view(B, T - 1, n), x[:, :1] - x0[:, None]), 1).reshape(B, -1)
    iq, iqc = dyn_res_ineq(x, u, x0, None, None, u_lower, u_upper)

    neq, nineq, nz = T * n, 2 * T * m, T * nt
    J = xu.new_zeros(B, neq + nineq, nz)
    Jeq = J[:, :neq].view(B, T, n, T, nt)
    ar = torch.arange(T - 1, device=xu.device)
    Jeq[:, ar, :, ar, :n] = -Jx.permute(1, 0, 2, 3)            # -df/dx_t
    Jeq[:, ar, :, ar, n:] = -Ju.permute(1, 0, 2, 3)            # -df/du_t
    eye_n = torch.eye(n, dtype=xu.dtype, device=xu.device)
    Jeq[:, ar, :, ar + 1, :n] = eye_n                          # +I on x_{t+1}
    Jeq[:, T - 1, :, 0, :n] = eye_n                            # x_0 - x0
    Jiq = J[:, neq:].view(B, T, 2, m, T, nt)
    at = torch.arange(T, device=xu.device)
    eye_m = torch.eye(m, dtype=xu.dtype, device=xu.device)
    Jiq[:, at, 0, :, at, n:] = eye_m
    Jiq[:, at, 1, :, at, n:] = -eye_m
    Jc = J.clone()
    Jc[:, neq:] *= (iqc > 0).to(xu.dtype)[..., None]
    return torch.cat((eq, iq), 1), torch.cat((eq, iqc), 1), J, Jc


# ---- the reference's own entry points into the Jacobian fill (al_utils.py:105-186,212-318), for callers
# that use them directly; the solver path goes through assemble_jacobian / the fused kernels
def constraint_res_jac2(xu, x0, dx_jac, x_lower, x_upper, u_lower, u_upper):
    """-> (res, res_clamp, constraint_jac, constraint_jac_clamp, constraint_hess) as al_utils.py:162-185."""
    res, resc, J, Jc = constraint_jacobian(xu, x0, dx_jac, u_lower, u_upper)
    return res, resc, J, Jc, torch.bmm(Jc.transpose(1, 2), Jc)


constraint_res_jac1 = constraint_res_jac2        # al_utils.py:140-160: the same quantities, older fill


def dyn_res_eq_jac(x, u, dx_jac, x0):
    """-> (res_eq (B, T n), its Jacobian (B, T n, T (n+m)))   (al_utils.py:212-262)."""
    n = x0.shape[-1]
    m = u.shape[-1]
    lo = torch.full((m,), -float("inf"), dtype=x.dtype, device=x.device)
    res, _, J, _ = constraint_jacobian(torch.cat((x, u), 2), x0, dx_jac, lo, -lo)
    neq = x.shape[1] * n
    return res[:, :neq], J[:, :neq]


def dyn_res_ineq_jac(x, u, x0, x_lower, x_upper, u_lower, u_upper):
    """-> (res, res_clamp, jac, jac_clamp) of the control-bound rows (al_utils.py:294-318)."""
    B, T, n = x.shape
    m = u.shape[-1]
    res, resc = dyn_res_ineq(x, u, x0, x_lower, x_upper, u_lower, u_upper)
    J = x.new_zeros(B, T, 2, m, T, n + m)
    at = torch.arange(T, device=x.device)
    eye_m = torch.eye(m, dtype=x.dtype, device=x.device)
    J[:, at, 0, :, at, n:] = eye_m
    J[:, at, 1, :, at, n:] = -eye_m
    J = J.reshape(B, 2 * T * m, T * (n + m))
    return res, resc, J, J * (resc > 0).to(x.dtype)[..., None]


def merit_hessian(xu, Q, q, dx_jac, x0, lamda, rho, x_lower, x_upper, u_lower, u_upper, diag_cost=True):
    """diag(Q) + rho Jc^T Jc, dense (B, nz, nz)   (al_utils.py:105-119)."""
    B = xu.shape[0]
    _, _, _, Jc = constraint_jacobian(xu, x0, dx_jac, u_lower, u_upper)
    Qfull = torch.diag_embed(Q.reshape(B, -1)) if diag_cost else Q
    return Qfull + rho.reshape(B, 1, 1) * torch.bmm(Jc.transpose(1, 2), Jc)


def compute_cost(xu, Q, q, diag_cost=True):
    """al_utils.py:339-351 (diagonal cost)."""
    assert diag_cost
    return (0.5 * (xu * Q * xu).sum(-1) + (q * xu).sum(-1)).sum(-1)


def compute_cost_gradient(xu, Q, q, diag_cost=True):
    """al_utils.py:354-360"""
    assert diag_cost
    return Q * xu + q


def merit_function(xu, Q, q, dx, x0, lamda, rho, x_lower, x_upper, u_lower, u_upper, diag_cost=True):
    """cost + rho/2 |res_clamp|^2 + lamda . res; a leading candidate axis (n_ls,B,T,nt) is
    folded into the batch (al_utils.py:37-59)."""
    m_ctrl = xu.shape[-1] - x0.shape[-1]
    if (xu.is_cuda and not torch.is_grad_enabled() and torch.is_tensor(u_lower) and u_lower.dim() == 1
            and torch.is_tensor(u_upper) and u_lower.numel() == m_ctrl and u_upper.numel() == m_ctrl
            and x_lower is None and x_upper is None):   # the kernel reads n_ctrl bounds per knot
        return _merit_fused(xu, Q, q, dx, x0, lamda, rho, u_lower, u_upper)
    if xu.dim() == 4:
        k, B = xu.shape[:2]
        rep = lambda t: t[None].expand(k, *t.shape).reshape(k * B, *t.shape[1:])
        xu, x0, Q, q, rho, lamda = xu.reshape(k * B, *xu.shape[2:]), rep(x0), rep(Q), rep(q), rep(rho), rep(lamda)
    B = xu.shape[0]
    res, resc = dyn_res(xu, dx, x0, x_lower, x_upper, u_lower, u_upper)
    return compute_cost(xu, Q, q) + 0.5 * rho[:, 0] * (resc * resc).sum(1) + (lamda * res).sum(1)


def _merit_fused(xu, Q, q, dx, x0, lamda, rho, u_lower, u_upper):
    """merit_function through dqp_al_merit: the dynamics are evaluated by the caller's module (one
    batched call over all candidates), cost + penalty + multiplier terms in one launch."""
    lib = _lib.load()
    cand = xu.dim() == 4
    k = xu.shape[0] if cand else 1
    flat = xu.reshape(-1, *xu.shape[-2:])
    n = x0.shape[-1]
    B, T, nt = x0.shape[0], flat.shape[1], flat.shape[2]
    m = nt - n
    x, u = _split(flat, n)
    if isinstance(dx, LinDx) and k > 1:     # F, f are per problem (time-major): tile over the candidates
        dx = LinDx(dx.F.repeat(1, k, 1, 1), dx.f.repeat(1, k, 1))
    x_next = _step(dx, x, u)
    d64 = lambda t: t.detach().double().contiguous()
    xu64, xn64 = d64(flat), d64(x_next)
    merit = torch.empty(k * B, dtype=torch.float64, device=xu.device)
    dims = _lib.dqp_al_mpc_dims(B, n, m, T)
    # the converted inputs must outlive the launch: a temporary freed right after data_ptr() is
    # handed back to the caching allocator and overwritten by the NEXT conversion before the
    # kernel (enqueued after all of them) reads it
    keep = [d64(x0), d64(Q), d64(q), d64(lamda), d64(rho).reshape(B), d64(u_lower), d64(u_upper)]
    with torch.cuda.device(xu.device):
        rc = lib.dqp_al_merit(ctypes.byref(dims), k, _ptr(xu64), _ptr(xn64), *[_ptr(t) for t in keep],
                              _ptr(merit), _stream(xu.device))
    _lib.check(rc, "dqp_al_merit")
    return merit.to(xu.dtype)


class HessianTerms:
    """diag(Qd) + rho Jc^T Jc, kept factored for the kernel (al_utils.py:96-102)."""

    def __init__(self, Jc, Qd, rho):
        self.Jc, self.Qd, self.rho = Jc, Qd, rho

    def dense(self):
        return torch.diag_embed(self.Qd) + self.rho[:, :, None] * torch.bmm(self.Jc.transpose(1, 2), self.Jc)


def merit_grad_hessian(xu, Q, q, dx, dx_jac, x0, lamda, rho, x_lower, x_upper, u_lower, u_upper,
                       diag_cost=True):
    """grad = Q xu + q + J^T lamda + rho Jc^T res_clamp ; Hessian terms (al_utils.py:62-102)."""
    B = xu.shape[0]
    Jc, gterm = assemble_jacobian(xu, x0, dx_jac, lamda, rho, u_lower, u_upper)
    grad = compute_cost_gradient(xu, Q, q, diag_cost).reshape(B, -1) + gterm.to(xu.dtype)
    return grad, HessianTerms(Jc, Q.reshape(B, -1), rho)


def assemble_jacobian(xu, x0, dx_jac, lamda, rho, u_lower, u_upper):
    """Clamped constraint Jacobian Jc (B,ncon,nz) and J^T lam + rho Jc^T res_c (B,nz) in one
    launch (dqp_al_assemble) from the per-knot dynamics Jacobians -- the fused form of
    constraint_jacobian() + the two bmm's the reference does (al_utils.py:62-102,162-318)."""
    lib = _lib.load()
    n = x0.shape[-1]
    B, T, nt = xu.shape
    m = nt - n
    x, u = _split(xu, n)
    x_next, (Jx, Ju) = dx_jac(x[:, :-1].reshape(-1, n), u[:, :-1].reshape(-1, m))
    eq = torch.cat((x[:, 1:] - x_next.view(B, T - 1, n), x[:, :1] - x0[:, None]), 1).reshape(B, -1)
    _, iqc = dyn_res_ineq(x, u, x0, None, None, u_lower, u_upper)
    resc = torch.cat((eq, iqc), 1).detach().double().contiguous()
    _need_gpu(resc)
    Jx = Jx.detach().double().reshape(B, T - 1, n, n).contiguous()
    Ju = Ju.detach().double().reshape(B, T - 1, n, m).contiguous()
    lam = lamda.detach().double().contiguous()
    rh = rho.detach().double().reshape(B).contiguous()
    dev = resc.device
    ncon, nz = T * n + 2 * T * m, T * nt
    Jc = torch.empty(B, ncon, nz, dtype=torch.float64, device=dev)
    gterm = torch.empty(B, nz, dtype=torch.float64, device=dev)
    dims = _lib.dqp_al_mpc_dims(B, n, m, T)
    with torch.cuda.device(dev):
        rc = lib.dqp_al_assemble(ctypes.byref(dims), _ptr(Jx), _ptr(Ju), _ptr(lam), _ptr(resc), _ptr(rh),
                                 _ptr(Jc), _ptr(gterm), _stream(dev))
    _lib.check(rc, "dqp_al_assemble")
    return Jc, gterm


# ------------------------------------------------------------------------------------------
# HIP calls
# ------------------------------------------------------------------------------------------
def newton_step(terms, grad):
    """-> (update = -H^-1 grad, L, info) through dqp_al_newton_step."""
    lib = _lib.load()
    Jc = terms.Jc.detach().double().contiguous()
    _need_gpu(Jc)
    B, ncon, nz = Jc.shape
    Qd = terms.Qd.detach().double().contiguous()
    rho = terms.rho.detach().double().reshape(B).contiguous()
    g = grad.detach().double().contiguous()
    dev = Jc.device
    upd = torch.empty(B, nz, dtype=torch.float64, device=dev)
    L = torch.empty(B, nz, nz, dtype=torch.float64, device=dev)
    info = torch.empty(B, dtype=torch.int32, device=dev)
    dims = _lib.dqp_al_dims(B, nz, ncon, 0)
    with torch.cuda.device(dev):
        rc = lib.dqp_al_newton_step(ctypes.byref(dims), _ptr(Jc), _ptr(Qd), _ptr(rho), _ptr(g),
                                    _ptr(upd), _ptr(L), _ptr(info), _stream(dev))
    _lib.check(rc, "dqp_al_newton_step")
    return upd, L, info


def chol_solve_neg(L, rhs):
    """-(L L^T)^-1 rhs through dqp_al_chol_solve."""
    lib = _lib.load()
    L = L.contiguous()
    _need_gpu(L)
    B, nz, _ = L.shape
    r = rhs.detach().double().reshape(B, nz).contiguous()
    out = torch.empty_like(r)
    dims = _lib.dqp_al_dims(B, nz, 0, 0)
    with torch.cuda.device(L.device):
        rc = lib.dqp_al_chol_solve(ctypes.byref(dims), _ptr(L), _ptr(r), _ptr(out), _stream(L.device))
    _lib.check(rc, "dqp_al_chol_solve")
    return out


# ------------------------------------------------------------------------------------------
def line_search_newton(update, x_est, meritfnQ, merit, x0):
    """20 candidate steps 2^-k evaluated as one batch; keep the argmin if it improves the merit
    (al_utils.py:503-527).  The x_0 entries of every candidate are pinned to x0."""
    B = x_est.shape[0]
    n = x0.shape[-1]
    steps = 2.0 ** (-torch.arange(N_LINESEARCH, device=x_est.device, dtype=torch.float32))
    steps = steps[:, None].expand(N_LINESEARCH, B)
    cand = x_est[None] + steps[:, :, None, None] * update[None]
    cand[:, :, 0, :n] = x0[None]
    vals = meritfnQ(cand).reshape(N_LINESEARCH, -1)
    best, idx = vals.min(dim=0)
    ar = torch.arange(B, device=x_est.device)
    x_new = cand[idx, ar]
    status = (best < merit).float()
    keep = status.to(x_est.dtype)[:, None, None]
    # mean accepted step: left on the device (the reference calls .item(), al_utils.py:527, a host
    # sync per Newton step; NewtonAL ignores the value)
    return keep * x_new + (1 - keep) * x_est, best, steps[idx, ar].mean(), status


DENSE_NEWTON_MAX_NZ = 128      # dqp_al_newton_step (csrc/dqp_al.hip): the Hessian of one problem in the LDS of a CU


class NewtonAL(torch.autograd.Function):
    """Four Newton steps on the augmented Lagrangian + implicit backward (al_utils.py:363-500)."""

    @staticmethod
    def forward(ctx, meritfn, dyn_fn, cost_fn, merit_grad_hessfn, xi, x0, lam, rho, Q, q,
                threshold, eps, ls):
        B, T, nt = xi.shape
        x_est = xi
        merit = meritfn(x_est, Q, q, lam, x0, rho)
        # beyond the dense Newton kernel's size (dqp_al_newton_step: nz <= 128) this general path is only reached as
        # the reference's LU fallback of a failed block-tridiagonal factorisation (AL_mpc.al_solve): solve by LU
        chol_failed = T * nt > DENSE_NEWTON_MAX_NZ
        status = None
        terms = L = None
        for _ in range(MAX_NEWTON_STEPS):          # merit_delta is pinned to 1000 (al_utils.py:453)
            with torch.enable_grad():   # autograd-based dx_jac callables need a leaf (al_utils.py:409-411)
                grad, terms = merit_grad_hessfn(x_est.detach().requires_grad_(True), Q, q, lam)
            grad = grad.detach()
            update = None
            if not chol_failed:
                upd, L, info = newton_step(terms, grad)
                update = upd.reshape(B, T, nt).to(x_est.dtype)
                # al_utils.py:419-423 scans `update` for NaN/Inf (two reductions + two host syncs);
                # the kernel reports the same condition per problem in `info` (NaN update <=> info != 0)
                if bool((info != 0).any()):
                    chol_failed = True
            if chol_failed:
                update = -torch.linalg.solve(terms.dense(), grad.reshape(B, -1)).reshape(B, T, nt)
            if ls:
                x_est, merit, _, status = line_search_newton(
                    update, x_est, lambda c: meritfn(c, Q, q, lam, x0, rho), merit, x0)
            else:
                x_est = x_est + update
                merit = meritfn(x_est, Q, q, lam, x0, rho)
        ctx.chol_failed = chol_failed
        ctx.terms = terms
        ctx.save_for_backward(L if L is not None else x_est.new_zeros(1), x_est)
        if status is None:
            status = torch.ones(B, device=xi.device)
        return x_est, status

    @staticmethod
    def backward(ctx, x_grad, status_grad):
        L, x = ctx.saved_tensors
        B = x_grad.shape[0]
        if ctx.chol_failed:
            g = -torch.linalg.solve(ctx.terms.dense(), x_grad.reshape(B, -1)).reshape(x_grad.shape)
        else:
            g = chol_solve_neg(L, x_grad).reshape(x_grad.shape).to(x_grad.dtype)
        # diagonal cost: dQ = g * x, dq = g                          al_utils.py:482-485
        return (None,) * 8 + (g * x, g, None, None, None)


def banded_jac_supported(B, n, m, T):
    """A block-tridiagonal Newton step exists for caller-linearised dynamics of these sizes."""
    dims = _lib.dqp_al_mpc_dims(B, n, m, T)
    return int(_lib.load().dqp_al_banded_factor_bytes(ctypes.byref(dims), 0)) > 0


class NewtonALBandedJac(torch.autograd.Function):
    """NewtonAL (al_utils.py:363-500) for a caller-supplied dynamics module at ANY horizon: the module's own
    linearisation (x_next, (Jx, Ju)) = dx_jac(x, u) goes to dqp_al_banded_newton_step_jac -- gradient,
    block-tridiagonal Hessian, block Cholesky and solve in one launch, no (B, ncon, nz) Jacobian and no
    (B, nz, nz) Hessian in memory (the dense step stops at nz = 128) -- and the line search evaluates the
    module on the 20 candidates as NewtonAL does."""

    @staticmethod
    def forward(ctx, meritfn, dx_jac, xi, x0, lam, rho, Q, q, u_lower, u_upper, ls):
        lib = _lib.load()
        B, T, nt = xi.shape
        n = x0.shape[-1]
        m = nt - n
        dev = xi.device
        kw = dict(dtype=torch.float64, device=dev)
        d64 = lambda t: t.detach().double().contiguous()
        dims = _lib.dqp_al_mpc_dims(B, n, m, T)
        rho_t = rho if torch.is_tensor(rho) else torch.full((B,), float(rho), **kw)
        keep = [d64(x0), d64(Q), d64(q), d64(lam), d64(rho_t).reshape(B), d64(u_lower).reshape(-1), d64(u_upper).reshape(-1)]
        fac = torch.empty(int(lib.dqp_al_banded_factor_bytes(ctypes.byref(dims), 0)) // 8, **kw)
        upd = torch.empty(B, T, nt, **kw)
        info = torch.empty(B, dtype=torch.int32, device=dev)
        x_est = xi
        merit = meritfn(x_est, Q, q, lam, x0, rho)
        status = None
        failed = False
        for _ in range(MAX_NEWTON_STEPS):
            xu = d64(x_est)
            with torch.enable_grad():       # autograd-based dx_jac callables need leaves (al_utils.py:409-411)
                xs = xu[:, :-1, :n].reshape(-1, n).to(xi.dtype).requires_grad_(True)
                us = xu[:, :-1, n:].reshape(-1, m).to(xi.dtype).requires_grad_(True)
                x_next, (Jx, Ju) = dx_jac(xs, us)
            hold = [d64(x_next), d64(Jx), d64(Ju)]
            with torch.cuda.device(dev):
                rc = lib.dqp_al_banded_newton_step_jac(ctypes.byref(dims), _ptr(xu), _ptr(keep[0]), _ptr(keep[1]), _ptr(keep[2]),
                                                       _ptr(keep[3]), _ptr(keep[4]), _ptr(keep[5]), _ptr(keep[6]),
                                                       _ptr(hold[0]), _ptr(hold[1]), _ptr(hold[2]), _ptr(upd), _ptr(fac),
                                                       _ptr(info), _stream(dev))
            _lib.check(rc, "dqp_al_banded_newton_step_jac")
            if bool((info != 0).any()):     # a pivot was not positive: the reference switches to an LU solve
                failed = True               # (al_utils.py:419-427); here the caller re-runs the dense path
                break
            update = upd.to(x_est.dtype)
            if ls:
                x_est, merit, _, status = line_search_newton(
                    update, x_est, lambda c: meritfn(c, Q, q, lam, x0, rho), merit, x0)
            else:
                x_est = x_est + update
                merit = meritfn(x_est, Q, q, lam, x0, rho)
        ctx.dims, ctx.failed = dims, failed
        ctx.save_for_backward(fac, x_est)
        if status is None:
            status = torch.ones(B, device=dev)
        return x_est, status, torch.tensor(failed, device=dev)

    @staticmethod
    def backward(ctx, x_grad, status_grad, failed_grad):
        fac, x = ctx.saved_tensors
        rhs = x_grad.detach().double().contiguous()
        g = torch.empty_like(rhs)
        with torch.cuda.device(rhs.device):
            rc = _lib.load().dqp_al_banded_solve(ctypes.byref(ctx.dims), 0, _ptr(fac), _ptr(rhs), _ptr(g), _stream(rhs.device))
        _lib.check(rc, "dqp_al_banded_solve")
        g = g.to(x_grad.dtype)
        # diagonal cost: dQ = g * x, dq = g                          al_utils.py:482-485
        return (None,) * 6 + (g * x, g, None, None, None)


class NewtonALDevice(torch.autograd.Function):
    """NewtonAL for a dynamics.DeviceDynamics: the four Newton steps as one C-ABI call
    (dqp_al_newton_solve: 21 launches, no host involvement); backward as NewtonAL's.  `slow` is a
    zero-argument callable that runs the general path (NewtonAL.apply with the Python closures): it
    is used when a Cholesky factorisation breaks down, where the reference switches the batch to an
    LU solve (al_utils.py:419-427)."""

    @staticmethod
    def forward(ctx, xi, x0, lam, rho, Q, q, dyn, u_lower, u_upper, slow, fail_sink=None):
        lib = _lib.load()
        B, T, nt = xi.shape
        n, m = dyn.n_state, dyn.n_ctrl
        dev = xi.device
        d64 = lambda t: t.detach().double().contiguous()
        xu = d64(xi).clone()
        keep = [d64(x0), d64(Q), d64(q), d64(lam), d64(rho).reshape(B), d64(u_lower).reshape(-1), d64(u_upper).reshape(-1)]
        dims = _lib.dqp_al_mpc_dims(B, n, m, T)
        kw = dict(dtype=torch.float64, device=dev)
        banded = 1 if BANDED_NEWTON_AL else 0
        if banded:      # block-tridiagonal factor, per knot (dqp_al_banded.hip)
            L = torch.empty(int(lib.dqp_al_banded_factor_bytes(ctypes.byref(dims), dyn.id)) // 8, **kw)
        else:
            L = torch.empty(B, T * nt, T * nt, **kw)
        status = torch.empty(B, **kw)
        fail = torch.zeros(1, dtype=torch.int32, device=dev)
        ws = torch.empty(int(lib.dqp_al_newton_solve_bytes(ctypes.byref(dims), banded)) // 8 + 1, **kw)
        with torch.cuda.device(dev):
            rc = lib.dqp_al_newton_solve(ctypes.byref(dims), dyn.id, dyn.dt, MAX_NEWTON_STEPS, banded,
                                         *[_ptr(t) for t in keep], _ptr(xu), _ptr(L), _ptr(status), _ptr(fail),
                                         _ptr(ws), _stream(dev))
        _lib.check(rc, "dqp_al_newton_solve")
        ctx.slow_ctx = None
        if fail_sink is not None:           # the caller checks all flags once, at the end of its solve
            fail_sink.append(fail)
        elif bool(fail.item()):             # rare: re-run through the general path, incl. its backward
            with torch.enable_grad():
                Qs, qs = Q.detach().requires_grad_(), q.detach().requires_grad_()
                out, st = slow(Qs, qs)
            ctx.slow_ctx = (out, Qs, qs)
            return out.detach(), st.detach()
        ctx.banded, ctx.dims, ctx.dyn_id = banded, dims, dyn.id
        ctx.save_for_backward(L, xu)
        return xu.to(xi.dtype), status.to(torch.float32)

    @staticmethod
    def backward(ctx, x_grad, status_grad):
        if ctx.slow_ctx is not None:
            out, Qs, qs = ctx.slow_ctx
            gQ, gq = torch.autograd.grad(out, (Qs, qs), x_grad)
            return (None,) * 4 + (gQ, gq) + (None,) * 5
        L, x = ctx.saved_tensors
        if ctx.banded:
            rhs = x_grad.detach().double().contiguous()
            g = torch.empty_like(rhs)
            with torch.cuda.device(rhs.device):
                rc = _lib.load().dqp_al_banded_solve(ctypes.byref(ctx.dims), ctx.dyn_id, _ptr(L), _ptr(rhs), _ptr(g),
                                                     _stream(rhs.device))
            _lib.check(rc, "dqp_al_banded_solve")
            g = g.to(x_grad.dtype)
        else:
            g = chol_solve_neg(L, x_grad).reshape(x_grad.shape).to(x_grad.dtype)
        return (None,) * 4 + (g * x.to(x_grad.dtype), g) + (None,) * 5      # al_utils.py:482-485


class ALSolveDevice(torch.autograd.Function):
    """AL_mpc.MPC.al_solve for a dynamics.DeviceDynamics as ONE C-ABI call (dqp_al_mpc_solve: start cost, warm
    start, al_iter x [four Newton steps with line search, multiplier / penalty update], no host involvement), and
    NewtonAL's backward through the last block-tridiagonal factor.  `prev` = (cost, lam, rho) history tensors of
    the previous call or None.  Returns xu (B,T,nt) and, non-differentiable, the new history (cost (K,B), lam
    (K,B,ncon), rho (K,B)), |res_clamp| (B) and the per-AL-iteration Cholesky-failure flags (int32)."""

    @staticmethod
    def forward(ctx, x_init, u_init, x0, Q, q, lam, rho, dyn, u_lower, u_upper, al_iter, prev):
        lib = _lib.load()
        B, T, n = x_init.shape
        m = u_init.shape[-1]
        nt, ncon = n + m, T * n + 2 * T * m
        dev = x0.device
        d64 = lambda t: t.detach().double().contiguous()
        keep = [d64(x_init), d64(u_init), d64(x0), d64(Q), d64(q), d64(u_lower).reshape(-1), d64(u_upper).reshape(-1),
                d64(lam), d64(rho).reshape(B)]
        _need_gpu(keep[0])
        pc, pl, pr = (d64(t) for t in prev) if prev is not None else (None, None, None)
        n_prev = pc.shape[0] if prev is not None else 0
        dims = _lib.dqp_al_mpc_dims(B, n, m, T)
        kw = dict(dtype=torch.float64, device=dev)
        xu = torch.empty(B, T, nt, **kw)
        hc, hl, hr = torch.empty(al_iter + 1, B, **kw), torch.empty(al_iter + 1, B, ncon, **kw), torch.empty(al_iter + 1, B, **kw)
        resn, status = torch.empty(B, **kw), torch.empty(B, **kw)
        L = torch.empty(int(lib.dqp_al_banded_factor_bytes(ctypes.byref(dims), dyn.id)) // 8, **kw)
        fail = torch.empty(al_iter, dtype=torch.int32, device=dev)
        ws = torch.empty(int(lib.dqp_al_mpc_solve_bytes(ctypes.byref(dims))) // 8 + 1, **kw)
        with torch.cuda.device(dev):
            rc = lib.dqp_al_mpc_solve(ctypes.byref(dims), dyn.id, dyn.dt, al_iter, MAX_NEWTON_STEPS, *[_ptr(t) for t in keep],
                                      _ptr(pc), _ptr(pl), _ptr(pr), n_prev, _ptr(xu), _ptr(hc), _ptr(hl), _ptr(hr), _ptr(resn),
                                      _ptr(L), _ptr(status), _ptr(fail), _ptr(ws), _stream(dev))
        _lib.check(rc, "dqp_al_mpc_solve")
        ctx.dims, ctx.dyn_id = dims, dyn.id
        ctx.save_for_backward(L, xu)
        ctx.mark_non_differentiable(hc, hl, hr, resn, fail)
        return xu, hc, hl, hr, resn, fail

    @staticmethod
    def backward(ctx, x_grad, *unused):
        L, x = ctx.saved_tensors
        rhs = x_grad.detach().double().contiguous()
        g = torch.empty_like(rhs)
        with torch.cuda.device(rhs.device):
            rc = _lib.load().dqp_al_banded_solve(ctypes.byref(ctx.dims), ctx.dyn_id, _ptr(L), _ptr(rhs), _ptr(g), _stream(rhs.device))
        _lib.check(rc, "dqp_al_banded_solve")
        return (None,) * 3 + (g * x, g) + (None,) * 7                      # al_utils.py:482-485


def outer_update_device(xu, x0, lam, rho, Q, q, dyn, u_lower, u_upper):
    """AL_mpc.py:296-307 in one launch (dqp_al_outer_update): -> (lam_new, cost (B), |res_clamp| (B))."""
    lib = _lib.load()
    B, T, nt = xu.shape
    dev = xu.device
    d64 = lambda t: t.detach().double().contiguous()
    keep = [d64(xu), d64(x0), d64(lam), d64(rho).reshape(B), d64(Q), d64(q), d64(u_lower).reshape(-1), d64(u_upper).reshape(-1)]
    kw = dict(dtype=torch.float64, device=dev)
    lam_new, cost, resn = torch.empty_like(keep[2]), torch.empty(B, **kw), torch.empty(B, **kw)
    dims = _lib.dqp_al_mpc_dims(B, dyn.n_state, dyn.n_ctrl, T)
    with torch.cuda.device(dev):
        rc = lib.dqp_al_outer_update(ctypes.byref(dims), dyn.id, dyn.dt, *[_ptr(t) for t in keep], _ptr(lam_new),
                                     _ptr(cost), _ptr(resn), _stream(dev))
    _lib.check(rc, "dqp_al_outer_update")
    return lam_new.to(lam.dtype), cost.to(xu.dtype), resn.to(xu.dtype)


def warm_start_al(x, lamda, rho, cost_start, cost_hist, lam_hist, rho_hist):
    """Pick the multipliers / penalty of the first stored AL iterate whose cost was already below
    the new starting cost; rescale lamda to that iterate's norm (al_utils.py:16-34)."""
    B = x.shape[0]
    idx = torch.max(cost_hist < cost_start[None], dim=0)[1]
    ar = torch.arange(B, device=x.device)
    ref = lam_hist[idx, ar]
    lamda = lamda * (ref.norm(p=2, dim=-1) / lamda.norm(p=2, dim=-1)).unsqueeze(-1)
    return lamda, rho_hist[idx, ar]
