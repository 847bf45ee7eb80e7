"""Device dynamics registry: the robots and pendulum modules the reference evaluates through
per-robot torch extensions (deqmpc/my_envs/{pendulum1l,cartpole1l,cartpole2l}, Python wrappers
deqmpc/my_envs/dynamics.py:15-263) or torch modules (deqmpc/envs.py:5-82,
qpth/env_dx/pendulum.py:18-83, deqmpc/rex_quadrotor.py:7-129), as HIP kernels behind the C ABI (include/dqp.h dqp_dyn_*).

    dyn = DeviceDynamics("cartpole1l", dt=0.05)
    x_next = dyn(x, u)                       # the `dx` callable of the MPC layers (differentiable)
    x_next, (Jx, Ju) = dyn.jac(x, u)         # the `dx_jac` callable (qp_wrapper.py:497,
                                             # al_utils.py:212-262): Jx (N,n,n), Ju (N,n,m)
    q_next, qdot_next = dyn.forward_dynamics(q, qdot, tau, h)      # the extension's own interface
    blocks = dyn.forward_derivatives(q, qdot, tau, h)              # (cartpole1l/src/dynamics_cpu.cpp:8-56)

The MPC layers recognise a DeviceDynamics and hand its id to the fused kernels instead of calling
back into Python.  There is no CPU path: tensors must live on the GPU.
"""
import ctypes

import torch

from . import _lib

NAMES = tuple(_lib.DQP_DYN)
DEFAULT_DT = {"pendulum1l": 0.05, "cartpole1l": 0.05, "cartpole2l": 0.05, "pendulum_euler": 0.05,
              "pendulum_dx": 0.05, "rexquadrotor": 0.05}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _gpu64(t, cols):
    if not t.is_cuda:
        raise RuntimeError("diff_qp_mpc_amd device dynamics run only on a GPU (HIP); got a %s tensor. "
                           "There is no CPU fallback." % t.device)
    t = t.detach()
    if t.dtype != torch.float64:
        t = t.double()
    t = t.reshape(-1, cols).contiguous()
    return t


class _Step(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, u, dyn):
        x64, u64 = _gpu64(x, dyn.n_state), _gpu64(u, dyn.n_ctrl)
        ctx.dyn = dyn
        ctx.shapes = (x.shape, u.shape, x.dtype, u.dtype)
        ctx.save_for_backward(x64, u64)
        return dyn._step(x64, u64).reshape(x.shape).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        x64, u64 = ctx.saved_tensors
        dyn = ctx.dyn
        xs, us, xd, ud = ctx.shapes
        _, Jx, Ju = dyn._jac(x64, u64, want_next=False)
        g64 = g.detach().double().reshape(-1, dyn.n_state, 1)
        gx = torch.bmm(Jx.transpose(1, 2), g64).reshape(xs).to(xd) if ctx.needs_input_grad[0] else None
        gu = torch.bmm(Ju.transpose(1, 2), g64).reshape(us).to(ud) if ctx.needs_input_grad[1] else None
        return gx, gu, None


class DeviceDynamics(torch.nn.Module):
    def __init__(self, name, dt=None):
        super().__init__()
        if name not in _lib.DQP_DYN:
            raise ValueError("unknown dynamics %r (registered: %s)" % (name, ", ".join(NAMES)))
        self.name = name
        self.id = _lib.DQP_DYN[name]
        self.dt = float(DEFAULT_DT[name] if dt is None else dt)
        n, m = ctypes.c_int32(0), ctypes.c_int32(0)
        _lib.check(_lib.load().dqp_dyn_sizes(self.id, ctypes.byref(n), ctypes.byref(m)), "dqp_dyn_sizes")
        self.n_state, self.n_ctrl = n.value, m.value
        self.nx, self.nu = self.n_state, self.n_ctrl
        self.nq = 7 if name == "rexquadrotor" else self.n_state // 2      # rex_quadrotor.py:163

    # ---- raw kernels on contiguous fp64 device tensors
    def _step(self, x64, u64):
        out = torch.empty_like(x64)
        with torch.cuda.device(x64.device):
            rc = _lib.load().dqp_dyn_step(self.id, x64.shape[0], _ptr(x64), _ptr(u64), self.dt, _ptr(out),
                                          _stream(x64.device))
        _lib.check(rc, "dqp_dyn_step")
        return out

    def _jac(self, x64, u64, want_next=True):
        N, n, m = x64.shape[0], self.n_state, self.n_ctrl
        xn = torch.empty_like(x64) if want_next else None
        Jx = torch.empty(N, n, n, dtype=torch.float64, device=x64.device)
        Ju = torch.empty(N, n, m, dtype=torch.float64, device=x64.device)
        with torch.cuda.device(x64.device):
            rc = _lib.load().dqp_dyn_jacobian(self.id, N, _ptr(x64), _ptr(u64), self.dt, _ptr(xn), _ptr(Jx),
                                              _ptr(Ju), _stream(x64.device))
        _lib.check(rc, "dqp_dyn_jacobian")
        return xn, Jx, Ju

    # ---- the callables the MPC layers take
    def forward(self, x, u):
        """x (..., n_state), u (..., n_ctrl) -> x_next, differentiable wrt both."""
        return _Step.apply(x, u, self)

    def jac(self, x, u):
        """-> (x_next, (Jx, Ju)), the reference's `dx_jac` contract (deqmpc/envs.py:74-82,
        deqmpc/my_envs/dynamics.py:253-263)."""
        lead = x.shape[:-1]
        xn, Jx, Ju = self._jac(_gpu64(x, self.n_state), _gpu64(u, self.n_ctrl))
        n, m = self.n_state, self.n_ctrl
        return (xn.reshape(*lead, n).to(x.dtype),
                (Jx.reshape(*lead, n, n).to(x.dtype), Ju.reshape(*lead, n, m).to(x.dtype)))

    dynamics_derivatives = jac

    # ---- the extension's interface (robots only)
    def forward_dynamics(self, q, qdot, tau, h):
        nq = self.nq
        q64, v64, t64 = _gpu64(q, nq), _gpu64(qdot, nq), _gpu64(tau, nq)
        h64 = _gpu64(h, 1).reshape(-1)
        qo, vo = torch.empty_like(q64), torch.empty_like(q64)
        with torch.cuda.device(q64.device):
            rc = _lib.load().dqp_dyn_forward_dynamics(self.id, q64.shape[0], _ptr(q64), _ptr(v64), _ptr(t64),
                                                      _ptr(h64), _ptr(qo), _ptr(vo), _stream(q64.device))
        _lib.check(rc, "dqp_dyn_forward_dynamics")
        return qo, vo

    def forward_derivatives(self, q, qdot, tau, h):
        nq = self.nq
        q64, v64, t64 = _gpu64(q, nq), _gpu64(qdot, nq), _gpu64(tau, nq)
        h64 = _gpu64(h, 1).reshape(-1)
        N = q64.shape[0]
        blocks = [torch.empty(N, nq, nq, dtype=torch.float64, device=q64.device) for _ in range(6)]
        with torch.cuda.device(q64.device):
            rc = _lib.load().dqp_dyn_forward_derivatives(self.id, N, _ptr(q64), _ptr(v64), _ptr(t64), _ptr(h64),
                                                         *[_ptr(b) for b in blocks], _stream(q64.device))
        _lib.check(rc, "dqp_dyn_forward_derivatives")
        return blocks


def recognise(module, n_state, n_ctrl, dt=None, device="cuda", samples=16, tol=1e-9):
    """Is `module` (a caller's torch dynamics, e.g. the reference's env.dynamics: deqmpc/envs.py PendulumDynamics,
    deqmpc/my_envs/dynamics.py Dynamics, rex_quadrotor.py) one of the registered device models?  Every registered
    model with these sizes is evaluated next to the module at `samples` seeded random points (step `dt`, default: the
    module's own `.dt` attribute, else the model's default) and the first one that reproduces it to `tol` (relative to
    1 + |x_next|) is returned as a DeviceDynamics -- so that a maintainer switching over keeps passing the env's module and
    still gets the on-chip solver paths.  Returns None when nothing matches (the module then takes the general paths:
    its own Jacobians into the block-tridiagonal Newton step, its residual into the caller-stepped PDIPM).  Nothing is
    substituted on a guess: the match is numerical, on the module the caller actually passed."""
    if isinstance(module, DeviceDynamics):
        return module
    gen = torch.Generator().manual_seed(0)
    x = (torch.rand(samples, n_state, generator=gen, dtype=torch.float64) * 2 - 1).to(device)
    u = (torch.rand(samples, n_ctrl, generator=gen, dtype=torch.float64) * 2 - 1).to(device)
    try:
        with torch.no_grad():
            ref = module(x, u)
    except Exception:
        return None
    if not torch.is_tensor(ref) or ref.shape != x.shape:
        return None
    step = dt if dt is not None else getattr(module, "dt", None)
    for name in NAMES:
        cand = DeviceDynamics(name, dt=float(step) if step is not None else None)
        if (cand.n_state, cand.n_ctrl) != (n_state, n_ctrl):
            continue
        got = cand(x, u)
        if bool(((got - ref.double()).abs() <= tol * (1.0 + ref.double().abs())).all()):
            return cand
    return None


class DynamicsResidual:
    """The `dyn_res` closure qp_wrapper.MPC hands to the QP solver (qp_wrapper.py:309,326-345):
    z (B, T (n+m)) -> [f(x_t,u_t) - x_{t+1}]_{t<T-1}, x_0 - x0 [, x_{T-1} (goal rows)], with f a
    registered device model.  QPFunction / DenseQPFunction recognise it and let the fused PDIPM
    evaluate it on chip every iteration (dqp_opts.dyn_*); calling it evaluates the same residual
    with the dqp_dyn_step kernel (used by the tests)."""

    def __init__(self, dynamics, x0, T, goal_rows=False):
        if not isinstance(dynamics, DeviceDynamics):
            raise TypeError("DynamicsResidual needs a DeviceDynamics")
        self.dynamics, self.T, self.goal_rows = dynamics, int(T), bool(goal_rows)
        self.x0 = _gpu64(x0, dynamics.n_state)

    def __call__(self, z):
        d = self.dynamics
        B = z.shape[0]
        tau = z.reshape(B, self.T, d.n_state + d.n_ctrl)
        xs, us = tau[..., :d.n_state], tau[..., d.n_state:]
        pred = d(xs[:, :-1].reshape(-1, d.n_state), us[:, :-1].reshape(-1, d.n_ctrl)).reshape(B, self.T - 1, d.n_state)
        parts = [(pred - xs[:, 1:]).reshape(B, -1), xs[:, 0] - self.x0.to(z.dtype)]
        if self.goal_rows:
            parts.append(xs[:, -1])
        return torch.cat(parts, dim=1)
