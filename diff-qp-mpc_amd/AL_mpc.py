"""AL_mpc.MPC with the reference's interface (qpth/AL_mpc.py:116-439): augmented-Lagrangian MPC,
`al_iter` outer multiplier/penalty updates around `al_utils.NewtonAL`.

    ctrl = MPC(n_state, n_ctrl, T, u_lower=..., u_upper=..., n_batch=B, u_init=..., dtype=torch.float64)
    ctrl.reinitialize(x0, mask)                       # required before the first forward (AL_mpc.py:432)
    x, u = ctrl(x0, QuadCost(C, c), dx, dx_jac)       # batch-major: C (B,T,nt,nt) c (B,T,nt)
                                                      # returns x (B,T,n), u (B,T,m) in float32

State carried across calls, as in the reference (AL_mpc.py:193-195,250-251,314-318): `lamda_prev`,
`rho_prev`, `cost_lam_hist` (the warm start looks up the stored AL iterate whose cost was already
below the new starting cost), `x_init`, `u_init`, `just_initialized`.
"""
import torch
from torch.nn import Module

from . import al_utils
from .al_utils import QuadCost, LinDx  # noqa: F401  (re-exported like the reference module)
from .dynamics import DeviceDynamics

# registered device models run NewtonAL's four Newton steps as one C-ABI call (no Python in between)
FUSED_NEWTON_AL = True
# registered device models with a block-tridiagonal kernel: the whole al_solve (warm start, al_iter x [Newton steps,
# multiplier update]) as one C-ABI call, dqp_al_mpc_solve; False: one call per Newton solve and outer update (round 2)
ONE_CALL_SOLVE = True
# one host synchronisation per al_solve to look at the Cholesky-failure flags of the device path
# (False: never look -- needed to capture a call in a hipGraph; a failed factorisation then leaves its
# problem at the last accepted iterate)
CHECK_CHOLESKY = True
# caller-supplied dynamics modules: the block-tridiagonal step on the module's own Jacobians (dqp_al_banded_newton_step_jac)
# at every size it exists for (n_state + n_ctrl <= 16, compiled pairs).  Round 3 retired the dense Newton step
# (dqp_al_newton_step: nz <= 128, 14 % of the fp64 peak with ~48 % useful FMAs in its cyclic LDL^T) as the default of
# this path: the banded step is faster at every size measured (tools/bench_al_user.py: nz 60: 9.7 -> 7.4 ms per call,
# nz 120: 19.8 -> 9.7 ms, same iterates to 1e-10) -- the dense kernel stays for shapes without a banded instantiation
# and behind this threshold (set it to 128 for round 2's behaviour)
BANDED_USER_DYNAMICS = True
BANDED_USER_DYNAMICS_FROM_NZ = 0


def _detach(t):
    if t is None or not torch.is_tensor(t):
        return t
    return t.detach() if t.requires_grad else t


class _DeviceHistory:
    """cost_lam_hist of a dqp_al_mpc_solve call: the reference's [[cost...], [lam...], [rho...]] lists (oldest first,
    AL_mpc.py:283,308-310) as views of three device tensors; indexes and iterates like the list of lists."""

    def __init__(self, cost, lam, rho):
        self.cost, self.lam, self.rho = cost, lam, rho

    def _lists(self):
        return [list(self.cost.unbind(0)), list(self.lam.unbind(0)), [r.unsqueeze(1) for r in self.rho.unbind(0)]]

    def __getitem__(self, i):
        return self._lists()[i]

    def __iter__(self):
        return iter(self._lists())

    def __len__(self):
        return 3


class MPC(Module):
    def __init__(self, n_state, n_ctrl, T, u_lower=None, u_upper=None, u_init=None, x_init=None,
                 al_iter=2, verbose=0, eps=1e-7, back_eps=1e-7, n_batch=None, linesearch_decay=0.2,
                 max_linesearch_iter=10, exit_unconverged=True, detach_unconverged=True,
                 backprop=True, slew_rate_penalty=None, solver_type='dense',
                 add_goal_constraint=False, x_goal=None, diag_cost=True, ineqG=None, ineqh=None,
                 dtype=torch.float64):
        super().__init__()
        assert (u_lower is None) == (u_upper is None)
        assert max_linesearch_iter > 0
        if u_lower is None:
            raise NotImplementedError("the reference's AL_mpc.MPC requires control bounds (AL_mpc.py:149-150)")
        if add_goal_constraint or ineqG is not None or not diag_cost:
            raise NotImplementedError("goal / general inequality constraints and dense costs are not "
                                      "reachable in the reference's al_solve either")
        self.dtype = dtype
        self.n_state, self.n_ctrl, self.T = n_state, n_ctrl, T
        self.u_lower = _detach(u_lower.to(dtype))
        self.u_upper = _detach(u_upper.to(dtype))
        self.x_lower = self.x_upper = None
        self.x_goal, self.ineqG, self.ineqh = x_goal, ineqG, ineqh
        self.u_init, self.x_init = _detach(u_init), _detach(x_init)
        self.verbose, self.eps, self.back_eps, self.n_batch = verbose, eps, back_eps, n_batch
        self.linesearch_decay, self.max_linesearch_iter = linesearch_decay, max_linesearch_iter
        self.exit_unconverged, self.detach_unconverged = exit_unconverged, detach_unconverged
        self.backprop, self.slew_rate_penalty, self.solver_type = backprop, slew_rate_penalty, solver_type
        self.add_goal_constraint, self.diag_cost, self.al_iter = add_goal_constraint, diag_cost, al_iter
        self.neq = n_state * T
        self.nineq = 2 * n_ctrl * T
        self.dyn_res_crit, self.dyn_res_factor = 1e-4, 10
        self.rho_prev = 1.0
        self.lamda_prev = torch.zeros(n_batch, self.neq + self.nineq, dtype=self.u_upper.dtype, device=self.u_upper.device)
        self.dyn_res_prev = 1000000
        self.fail_log = []
        self.mask = torch.ones(n_batch, T, 1, dtype=self.u_upper.dtype, device=self.u_upper.device)

    # ------------------------------------------------------------------ AL_mpc.py:198-252
    def forward(self, x0, cost, dx, dx_jac, u_init=None, x_init=None):
        B = self.n_batch if self.n_batch is not None else cost.C.size(0)
        assert cost.C.ndimension() == 4
        assert x0.ndimension() == 2 and x0.size(0) == B

        def batched(t):
            return t.unsqueeze(0).expand(B, self.T, -1).clone() if t.ndimension() == 2 else t

        if u_init is not None:
            u = batched(u_init)
        elif self.u_init is None:
            u = torch.zeros(B, self.T, self.n_ctrl, dtype=x0.dtype, device=x0.device)
        else:
            u = batched(self.u_init)
        u = u.type_as(x0.data)
        if x_init is not None:
            x = batched(x_init)
        elif self.x_init is None:
            x = self.rollout(x0, u, dx)
        else:
            x = batched(self.x_init)
        x = x.type_as(x0.data)
        if self.diag_cost:
            cost = QuadCost(cost.C.diagonal(dim1=-2, dim2=-1), cost.c)
        x, u = self.al_solve(x, u, dx, dx_jac, x0, cost)
        self.x_init, self.u_init = x, u
        return (x, u)

    # ------------------------------------------------------------------ AL_mpc.py:254-321
    def al_solve(self, x, u, dx, dx_jac, x0, cost, lamda_init=None, rho_init=None, _fused=True):
        dt = self.dtype
        x_in, u_in, x0_in = x, u, x0
        nt = self.n_state + self.n_ctrl
        banded = al_utils.BANDED_NEWTON_AL and nt <= 16                # one knot per 16-lane DPP row
        dense = self.n_state <= 8 and self.n_ctrl <= 2 and self.T * nt <= 128
        device_path = (_fused and FUSED_NEWTON_AL and isinstance(dx, DeviceDynamics) and (banded or dense)
                       and self.x_lower is None and self.u_lower.numel() == self.n_ctrl)
        if (ONE_CALL_SOLVE and device_path and banded and al_utils.BANDED_NEWTON_AL and torch.is_tensor(self.rho_prev if rho_init is None else rho_init)
                and dt == torch.float64):
            # the whole solve -- start cost, warm start, al_iter x [Newton steps + line search, multiplier update] -- as one
            # C-ABI call (dqp_al_mpc_solve); the history stays on the device as three tensors
            lamda = self.lamda_prev.to(dt) if lamda_init is None else lamda_init
            rho = self.rho_prev if rho_init is None else rho_init
            prev = None if self.just_initialized else self._device_history()
            xu, hc, hl, hr, resn, fail = al_utils.ALSolveDevice.apply(          # the iterate enters detached (AL_mpc.py:287)
                x.detach(), u.detach(), x0.detach(), cost.C.to(dt), cost.c.to(dt), lamda.detach(), rho.detach(), dx,
                self.u_lower, self.u_upper, self.al_iter, prev)
            self.fail_log.append(fail)          # per-AL-iteration Cholesky-failure flags of the calls since reinitialize()
            if CHECK_CHOLESKY and bool(fail.any()):
                # a Cholesky factorisation broke down somewhere in the batch: the reference then switches the batch to an
                # LU solve (al_utils.py:419-427) -- redo this solve on the general path
                return self.al_solve(x_in, u_in, dx, dx_jac, x0_in, cost, lamda_init, rho_init, _fused=False)
            self.cost_lam_hist = _DeviceHistory(hc, hl, hr)
            self.lamda_prev, self.rho_prev, self.dyn_res_prev = hl[-1], hr[-1].unsqueeze(1), resn
            self.just_initialized = False
            return xu[:, :, :self.n_state].float(), xu[:, :, self.n_state:].float()                  # AL_mpc.py:319-320
        fail_flags = []
        x, u, x0 = x.to(dt), u.to(dt), x0.to(dt)
        lamda = self.lamda_prev.to(dt) if lamda_init is None else lamda_init
        rho = self.rho_prev if rho_init is None else rho_init
        with torch.no_grad():
            xu0 = torch.cat((x, u), dim=2)
            cost_start = self.compute_cost(xu0, cost.C.double(), cost.c.double())
            if not self.just_initialized:
                hist = [torch.stack(list(h)[::-1], dim=0) for h in self.cost_lam_hist]
                lamda, rho = al_utils.warm_start_al(x, lamda, rho, cost_start, *hist)
        Q, q = cost.C.to(dt), cost.c.to(dt)
        history = [[cost_start], [lamda], [rho]]
        for _ in range(self.al_iter):
            xu = torch.cat((x, u), dim=2).detach().clone()
            rho_i = rho
            def general(Qg, qg, xu=xu, lamda=lamda, rho=rho, rho_i=rho_i):
                return al_utils.NewtonAL.apply(
                    lambda xi, Qi, qi, yi, x0i=x0, rhoi=rho_i, grad=False:
                        self.merit_function(xi, Qi, qi, dx, x0i, yi, rhoi, grad),
                    lambda xi: self.dyn_res(xi, dx, x0),
                    lambda xi, Qi, qi: self.compute_cost(xi, Qi, qi),
                    lambda xi, Qi, qi, yi: self.merit_grad_hess(xi, Qi, qi, dx, dx_jac, x0, yi, rho_i),
                    xu, x0, lamda, rho, Qg, qg, 1e-3, 1e-6, True)
            banded_jac = (not device_path and _fused and BANDED_USER_DYNAMICS and x.is_cuda and dx_jac is not None
                          and self.x_lower is None and self.u_lower.numel() == self.n_ctrl
                          and self.T * nt > BANDED_USER_DYNAMICS_FROM_NZ
                          and al_utils.banded_jac_supported(self.n_batch, self.n_state, self.n_ctrl, self.T))
            if banded_jac:
                # caller-supplied dynamics at a horizon the dense Newton step cannot hold (nz > 128): its own
                # Jacobians into the block-tridiagonal step
                out, status, failed = al_utils.NewtonALBandedJac.apply(
                    lambda xi, Qi, qi, yi, x0i=x0, rhoi=rho_i, grad=False:
                        self.merit_function(xi, Qi, qi, dx, x0i, yi, rhoi, grad),
                    dx_jac, xu, x0, lamda, rho, Q, q, self.u_lower, self.u_upper, True)
                fail_flags.append(failed.reshape(1).to(torch.int32))
            elif device_path and torch.is_tensor(rho):
                # registered device model: the four Newton steps in one C-ABI call (dqp_al_newton_solve);
                # Cholesky-failure flags are collected and checked once after the last AL iteration
                out, status = al_utils.NewtonALDevice.apply(xu, x0, lamda, rho, Q, q, dx, self.u_lower,
                                                            self.u_upper, general, fail_flags)
            else:
                out, status = general(Q, q)
            x, u = out[:, :, :self.n_state], out[:, :, self.n_state:]
            with torch.no_grad():
                if device_path and torch.is_tensor(rho):
                    lamda, cost_res, dyn_res_clamp = al_utils.outer_update_device(
                        out, x0, lamda, rho, Q, q, dx, self.u_lower, self.u_upper)
                else:
                    res, res_clamp = self.dyn_res(torch.cat((x, u), dim=2), dx, x0, res_type='both')
                    lamda = lamda + rho * res                                 # AL_mpc.py:299
                    lamda = torch.cat([lamda[:, :self.neq], lamda[:, self.neq:].clamp(min=0)], dim=1)
                    cost_res = self.compute_cost(out, Q, q)
                    dyn_res_clamp = res_clamp.view(self.n_batch, -1).norm(dim=-1)
                rho = rho * 10                                             # AL_mpc.py:307
                history[0].append(cost_res)
                history[1].append(lamda)
                history[2].append(rho)
        if fail_flags and CHECK_CHOLESKY and bool(torch.stack(fail_flags).any()):
            # a Cholesky factorisation broke down somewhere in the batch: the reference then switches
            # the batch to an LU solve (al_utils.py:419-427) -- redo this solve on the general path
            return self.al_solve(x_in, u_in, dx, dx_jac, x0_in, cost, lamda_init, rho_init, _fused=False)
        self.cost_lam_hist = history
        self.lamda_prev, self.rho_prev, self.dyn_res_prev = lamda, rho, dyn_res_clamp
        self.just_initialized = False
        return x.float(), u.float()                                        # AL_mpc.py:319-320

    def _device_history(self):
        """cost_lam_hist as the three device tensors dqp_al_mpc_solve reads: (K,B), (K,B,ncon), (K,B), oldest first"""
        h = self.cost_lam_hist
        if isinstance(h, _DeviceHistory):
            return h.cost, h.lam, h.rho
        B = self.n_batch
        return (torch.stack([c.reshape(B) for c in h[0]]), torch.stack(list(h[1])),
                torch.stack([r.reshape(B) for r in h[2]]))

    # ------------------------------------------------------------------ thin wrappers (AL_mpc.py:323-429)
    def merit_function(self, xu, Q, q, dx, x0, lamda, rho, grad=False):
        return al_utils.merit_function(xu, Q, q, dx, x0, lamda, rho, self.x_lower, self.x_upper,
                                       self.u_lower, self.u_upper, self.diag_cost)

    def merit_grad_hess(self, xu, Q, q, dx, dx_jac, x0, lamda, rho):
        return al_utils.merit_grad_hessian(xu, Q, q, dx, dx_jac, x0, lamda, rho, self.x_lower,
                                           self.x_upper, self.u_lower, self.u_upper, self.diag_cost)

    def merit_hessian(self, xu, Q, q, dx, dx_jac, x0, lamda, rho):
        """AL_mpc.py:325-326: the dense merit Hessian diag(Q) + rho Jc^T Jc (B, nz, nz)."""
        _, terms = self.merit_grad_hess(xu, Q, q, dx, dx_jac, x0, lamda, rho)
        return terms.dense()

    def dyn_res_eq(self, x, u, dx, x0, mask=None):
        """AL_mpc.py:330-355 (the mask is unused there too)."""
        return al_utils.dyn_res_eq(x, u, dx, x0)

    def dyn_res_ineq(self, x, u, dx, x0):
        """AL_mpc.py:357-383: (res, res_clamp) of the control bounds."""
        return al_utils.dyn_res_ineq(x, u, x0, self.x_lower, self.x_upper, self.u_lower, self.u_upper)

    def rollout_lin(self, x, actions, F, f):
        """AL_mpc.py:414-424: batch-major rollout under time-varying linear dynamics."""
        return self.rollout(x, actions, LinDx(F, f))          # this class's rollout indexes LinDx batch-major

    def dyn_res(self, xu, dx, x0, res_type='clamp'):
        res, res_clamp = al_utils.dyn_res(xu, dx, x0, self.x_lower, self.x_upper, self.u_lower, self.u_upper)
        if res_type == 'noclamp':
            return res
        if res_type == 'clamp':
            return res_clamp
        return res, res_clamp

    def rollout(self, x, actions, dynamics):
        if isinstance(dynamics, DeviceDynamics) and x.is_cuda and self.n_state <= 12 and self.n_ctrl <= 8:
            # one launch (and one for its adjoint) instead of T-1 dynamics calls
            from .qp_wrapper import _Rollout
            xs = _Rollout.apply(x, actions.transpose(0, 1), None, None, dynamics, self.n_state, self.n_ctrl, self.T)
            return xs.transpose(0, 1)
        xs = [x]
        for t in range(self.T - 1):
            xt, ut = xs[t], actions[:, t]
            if isinstance(dynamics, LinDx):
                nxt = torch.bmm(dynamics.F[:, t], torch.cat([xt, ut], dim=-1).unsqueeze(2)).squeeze(2) \
                    + dynamics.f[:, t]
            else:
                nxt = dynamics(xt, ut)
            xs.append(nxt)
        return torch.stack(xs, 1)

    def compute_cost(self, xu, Q, q):
        return al_utils.compute_cost(xu, Q, q, self.diag_cost)

    def compute_cost_gradient(self, xu, Q, q):
        return al_utils.compute_cost_gradient(xu, Q, q, self.diag_cost)

    def reinitialize(self, x, mask):
        self.u_init = self.x_init = None
        self.rho_prev = torch.ones((self.n_batch, 1), device=x.device, dtype=x.dtype)
        self.lamda_prev = torch.zeros(self.n_batch, self.neq + self.nineq, device=x.device, dtype=x.dtype)
        self.dyn_res_prev = 1000000
        self.just_initialized = True
        self.mask = mask
        self.fail_log = []


class GraphedMPC:
    """One COLD call of an AL_mpc.MPC -- reinitialize() + forward -- on a registered device model captured as two
    hipGraphs (forward; backward through NewtonAL's implicit derivative), like qp_wrapper.GraphedMPC: the solve is C-ABI
    calls that only enqueue on the current stream.  Shapes and the batch size are frozen at capture; inputs are copied into
    the graph's static buffers.  The Cholesky-failure flags are not looked at inside the graph (CHECK_CHOLESKY is off for
    the captured call): `failed()` reads those of the last replay.

        g = GraphedMPC(ctrl, (x0, C, c), dyn, x_init=x_ref, u_init=u_ref);   x, u = g(x0, C, c)
    """

    def __init__(self, ctrl, sample_inputs, dyn, x_init=None, u_init=None, warmup=3):
        global CHECK_CHOLESKY
        from .qp_wrapper import _GraphReplay
        self._replay = _GraphReplay
        self.ctrl, self.dyn = ctrl, dyn
        self.x_init, self.u_init = x_init, u_init
        self.static_inputs = tuple(t.detach().clone().requires_grad_(t.requires_grad) for t in sample_inputs)
        check, CHECK_CHOLESKY = CHECK_CHOLESKY, False
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    outs = self._call(*self.static_inputs)
                    torch.autograd.grad(outs, self._grad_inputs(), tuple(torch.ones_like(o) for o in outs), allow_unused=True)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            pool = torch.cuda.graph_pool_handle()
            self.fwd_graph, self.bwd_graph = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.fwd_graph, pool=pool):
                self.static_outputs = self._call(*self.static_inputs)
                self._flags = list(ctrl.fail_log)
            self.static_grad_outputs = tuple(torch.zeros_like(o) for o in self.static_outputs)
            with torch.cuda.graph(self.bwd_graph, pool=pool):
                gi = torch.autograd.grad(self.static_outputs, self._grad_inputs(), self.static_grad_outputs, allow_unused=True)
        finally:
            CHECK_CHOLESKY = check
        it = iter(gi)
        self.static_grad_inputs = tuple(next(it) if t.requires_grad else None for t in self.static_inputs)

    def _grad_inputs(self):
        return tuple(t for t in self.static_inputs if t.requires_grad)

    def _call(self, x0, C, c):
        ctrl = self.ctrl
        ctrl.reinitialize(x0, ctrl.mask)
        ctrl.x_init, ctrl.u_init = self.x_init, self.u_init
        return ctrl(x0, QuadCost(C, c), self.dyn, self.dyn.jac)

    def __call__(self, *inputs):
        return self._replay.apply(self, *inputs)

    def failed(self):
        return any(bool(f.any()) for f in self._flags)
