"""sl1qp_mpc.MPC with the reference's interface (qpth/sl1qp_mpc.py:59-330): qp_wrapper.MPC whose QP is solved in the
l1-slack ("SL1QP") form of sl1qpify (sl1qp_mpc.py:703-752) -- dynamics and bound violations are allowed at a price
`mu` per unit, which keeps the QP feasible for any linearisation.

    ctrl = MPC(n_state, n_ctrl, T, u_lower=..., u_upper=..., mu=..., qp_iter=..., single_qp_solve=...)
    x, u = ctrl(x0, QuadCost(C, c), dx[, dx_true])            # note: no dx_jac argument (sl1qp_mpc.py:216)

Differences from the reference's file, which cannot run as shipped: its single_qp stops at an
unconditional ipdb.set_trace() (sl1qp_mpc.py:326), its sl1qpify sizes the equality slacks with the inequality count
(it only type-checks for neq = nineq) and leaves the slack block of Q exactly zero, which its own PDIPM cannot factor.
Here the formulation is sl1qp.sl1qpify (right block shapes, `reg` I on the slack block), the extended QP -- nz + 2 neq +
nineq variables: 90 at n 3 m 3 T 5, 120 at the pendulum's T 10 -- runs on the blocked dense kernels (csrc/dqp_big.hip)
through DenseQPFunction, and the equality residual inside the iterations is the extended linear form
A z - v + w - b (the reference passes the true-dynamics residual minus v plus w, sl1qp_mpc.py:344-364: identical for
LinDx; for a nonlinear model this is the `linearised_residual` behaviour of qp_wrapper.MPC).  Parity of the clone is
unpinned (nothing to run in the reference); the formulation is checked against the CPU restatement (tests only) on the extended QP and
against qp_wrapper.MPC in the exact-penalty regime (tests/test_gpu_big.py).
"""
import torch

from . import qp_wrapper
from .qp import DenseQPFunction
from .qp_wrapper import GradMethods, LinDx, QuadCost, detach_maybe  # noqa: F401  (re-exported like the reference module)
from .dynamics import DeviceDynamics
from .sl1qp import sl1qpify


def _autograd_jac(dx):
    """(x_next, (df/dx, df/du)) of a torch dynamics module by reverse-mode autograd, one sample at a time batched with
    vmap -- what the reference's linearize_dynamics derives when it is handed no Jacobian function (sl1qp_mpc.py:216)."""
    def jac(xs, us):
        fx, fu = torch.func.vmap(torch.func.jacrev(lambda a, b: dx(a[None], b[None])[0], argnums=(0, 1)))(xs, us)
        return dx(xs, us), (fx, fu)
    return jac


class MPC(qp_wrapper.MPC):
    """Constructor arguments as qp_wrapper.MPC plus `mu` (default 1) and `reg`; max_linesearch_iter defaults to 1 as in
    the reference (sl1qp_mpc.py:139)."""

    def __init__(self, n_state, n_ctrl, T, u_lower=None, u_upper=None, u_zero_I=None, u_init=None, x_init=None,
                 qp_iter=10, grad_method=GradMethods.ANALYTIC, delta_u=None, verbose=0, eps=1e-7, back_eps=1e-7, mu=1,
                 n_batch=None, linesearch_decay=0.2, max_linesearch_iter=1, exit_unconverged=True,
                 detach_unconverged=True, backprop=True, slew_rate_penalty=None, prev_ctrl=None, not_improved_lim=5,
                 best_cost_eps=1e-4, solver_type='dense', single_qp_solve=False, add_goal_constraint=False, x_goal=None,
                 reg=1e-6):
        if add_goal_constraint:
            raise NotImplementedError("add_goal_constraint: the reference's sl1qp clone sizes its slack blocks for the "
                                      "dynamics rows only (sl1qp_mpc.py:723-751)")
        if u_lower is None:
            raise NotImplementedError("the l1-slack form needs control bounds (its G block, sl1qp_mpc.py:693-700)")
        super().__init__(n_state, n_ctrl, T, u_lower=u_lower, u_upper=u_upper, u_zero_I=u_zero_I, u_init=u_init,
                         x_init=x_init, qp_iter=qp_iter, grad_method=grad_method, delta_u=delta_u, verbose=verbose,
                         eps=eps, back_eps=back_eps, n_batch=n_batch, linesearch_decay=linesearch_decay,
                         max_linesearch_iter=max_linesearch_iter, exit_unconverged=exit_unconverged,
                         detach_unconverged=detach_unconverged, backprop=backprop, slew_rate_penalty=slew_rate_penalty,
                         prev_ctrl=prev_ctrl, not_improved_lim=not_improved_lim, best_cost_eps=best_cost_eps,
                         solver_type=solver_type, single_qp_solve=single_qp_solve, add_goal_constraint=False,
                         x_goal=x_goal, linearised_residual=True)
        self.mu, self.reg = mu, reg
        self.xu_dim = T * (n_state + n_ctrl)                    # sl1qp_mpc.py:194
        self.num_eq, self.num_ineq = T * n_state, 2 * T * n_ctrl
        self.slacks = None

    def forward(self, x0, cost, dx, dx_true=None):
        if isinstance(dx, LinDx):
            dx_jac = None
        elif isinstance(dx, DeviceDynamics):
            dx_jac = dx.jac
        else:
            dx_jac = _autograd_jac(dx)
        return super().forward(x0, cost, dx, dx_jac, dx_true)

    def single_qp(self, x, u, dx, dx_jac, x0, cost, need_cost=True):
        """sl1qp_mpc.py:298-330: linearise, assemble the dense MPC QP, soften it (sl1qpify), solve; the slacks
        (v, w, t) of the solution are kept on `self.slacks`."""
        if isinstance(dx, LinDx):
            F = dx.F
            f = dx.f if dx.f is not None else torch.zeros(self.T - 1, self.n_batch, self.n_state, dtype=x0.dtype, device=x0.device)
        else:
            F, f = self.linearize_dynamics(x, detach_maybe(u), dx, dx_jac, diff=False)
        Q, q, G, h, A, b = self._dense(cost.C, cost.c, F, f, x0)
        ext = sl1qpify(Q, q, G, h, A, b, self.mu, self.reg)
        sol = DenseQPFunction(verbose=-1)(*ext).to(x0.dtype)
        nz, ne, ni = self.xu_dim, self.num_eq, self.num_ineq
        self.slacks = (sol[:, nz:nz + ne].detach(), sol[:, nz + ne:nz + 2 * ne].detach(), sol[:, nz + 2 * ne:nz + 2 * ne + ni].detach())
        tau = sol[:, :nz].reshape(self.n_batch, self.T, -1)
        x_qp, u_qp = tau[..., :self.n_state].transpose(0, 1), tau[..., self.n_state:].transpose(0, 1)
        return x_qp - x, u_qp - u, (self.compute_cost(tau, cost) if need_cost else None)
