"""DEQ-MPC call pattern: the solver-facing part of deqmpc/policies.py, on the
fused kernels.

    DEQLayer       policies.py:191-430   the MLP variant (layer_type "mlp"; state-dict keys as the
                                         reference's, so its checkpoints load)
    Tracking_MPC   policies.py:567-686   p = -(Q x_ref); AL_mpc.MPC ("al", the reference's default,
                                         deqmpc/train.py:61) or qp_wrapper.MPC ("ip")
    DEQMPCPolicy   policies.py:432-529   deq_iter x [DEQLayer -> Tracking_MPC]; every iterate is returned
    FFDNetwork / NNMPCPolicy   policies.py:532-716   feed-forward reference + the same Tracking_MPC
    NNPolicy       policies.py:719-784   behaviour-cloning baseline (no solver)
    compute_loss / compute_loss_deqmpc / _deq / _bc / add_loss_based_on_out_type   policies.py:787-848
    (DEQPolicy, policies.py:25-128, is dead code in the reference: its solver is an unbound `anderson` using the
    removed torch.solve and an undefined self.kwargs; not mirrored)
    train_step     the body of deqmpc/train.py:135-175 for one batch, plus what the reference does not
                   have: data-parallel training -- each rank solves its own shard of trajectories
                   (the solves never communicate) and the DEQLayer gradients are summed over ranks with
                   ONE flat all_reduce (RCCL) before the optimiser step.

The dynamics are passed as in the reference (`env.dynamics`, `env.dynamics_derivatives`); a
dynamics.DeviceDynamics (its `.jac` is the derivatives callable) keeps every solver call on the GPU.
"""
import torch
import torch.nn as nn

from . import AL_mpc as al_mpc
from .dynamics import DeviceDynamics, recognise
from . import al_utils
from . import qp_wrapper as ip_mpc


# Tracking_MPC: look at the env's dynamics module and use the registered device model when it is one (dynamics.recognise)
RECOGNISE_ENV_DYNAMICS = True


# batch rows from which the DEQLayer's linears take their weight gradient in slices (SlicedLinear below)
SLICED_WGRAD_FROM = 8192


class SlicedLinear(torch.autograd.Function):
    """y = x W^T + b whose weight gradient dW = dY^T X is formed in row slices.  At the reference's training sizes the
    batch (the GEMM's reduction length) is 10^4 - 10^5 rows against a 128 x 128 result: the library heuristic runs that
    as ONE pass of sixteen 32 x 32 tiles over all rows (226 us per call at B = 65536 on MI355X, 11 % of a config-5
    training step, profiles/r3/config5_kernel_stats.csv); as a batched product over S row slices plus a sum it fills
    the chip.  Forward and dX are the plain library calls."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return torch.nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        rows = x.shape[0]
        s = 1
        while rows % (2 * s) == 0 and rows // (2 * s) >= 512:
            s *= 2
        gx = gy @ w if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            gw = torch.bmm(gy.view(s, rows // s, -1).transpose(1, 2), x.contiguous().view(s, rows // s, -1)).sum(0)
        gb = gy.sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return gx, gw, gb


def _linear(lin, x):
    if x.is_cuda and x.dim() == 2 and x.shape[0] >= SLICED_WGRAD_FROM and torch.is_grad_enabled():
        return SlicedLinear.apply(x, lin.weight, lin.bias)
    return lin(x)


class DEQLayer(nn.Module):
    """One DEQ iteration of the reference's trajectory network, MLP flavour (policies.py:191-430,
    layer_type "mlp", deq_out_type 1 or 2): input = current reference trajectory, hidden state z."""

    def __init__(self, args, env):
        super().__init__()
        self.args = args
        self.nu, self.nx, self.nq = env.nu, env.nx, args.nq
        self.dt, self.T, self.hdim = env.dt, args.T, args.hdim
        self.layer_type = getattr(args, "layer_type", "mlp")
        self.out_type = args.deq_out_type
        if self.layer_type != "mlp":
            # the reference's other variants do not construct: "gcn" reads self.num_groups, which nothing sets
            # (policies.py:386-390 -> AttributeError), and chains Linear(hdim, 4 hdim) into Linear(3 hdim, hdim)
            # (:352-356); "gat" is a bare `NotImplementedError` expression in every branch (:264,285,299,...)
            raise NotImplementedError("layer_type %r: only the MLP DEQLayer exists in working form in the reference "
                                      "(its gcn variant fails in __init__, gat is a stub)" % self.layer_type)
        if self.out_type not in (1, 2):
            raise NotImplementedError("deq_out_type 1 / 2 (state prediction), as DEQMPCPolicy.forward handles")
        self.in_dim = self.nx + self.nx * (self.T - 1)                             # policies.py:313-315
        self.inp_layer = nn.Sequential(nn.Linear(self.in_dim, self.hdim), nn.LayerNorm(self.hdim))
        self.fcdeq1, self.lndeq1, self.reludeq1 = nn.Linear(self.hdim, self.hdim), nn.LayerNorm(self.hdim), nn.ReLU()
        self.fcdeq2, self.lndeq2, self.reludeq2 = nn.Linear(self.hdim, self.hdim), nn.LayerNorm(self.hdim), nn.ReLU()
        self.lndeq3 = nn.LayerNorm(self.hdim)
        self.out_dim = self.nx * (self.T - 1) if self.out_type == 1 else self.nx * self.T
        self.out_layer = nn.Sequential(nn.Linear(self.hdim, self.out_dim))

    def init_z(self, bsz):
        p = self.fcdeq1.weight
        return torch.zeros(bsz, self.hdim, dtype=torch.float32, device=p.device)

    def deq_layer(self, x, z):                                                    # policies.py:277-283
        z = self.lndeq1(self.reludeq1(_linear(self.fcdeq1, z)))
        return self.lndeq3(self.reludeq2(z + self.lndeq2(x + _linear(self.fcdeq2, z))))

    def forward(self, x, z):
        z_out = self.deq_layer(self.inp_layer[1](_linear(self.inp_layer[0], x)), z)
        steps = self.T - 1 if self.out_type == 1 else self.T
        dx_ref = _linear(self.out_layer[0], z_out).view(-1, steps, self.nx)
        vel_ref = dx_ref[..., self.nq:]
        pos_ref = dx_ref[..., :self.nq] * self.dt + x[:, None, :self.nq]           # policies.py:221-223
        return torch.cat([pos_ref, vel_ref], dim=-1), z_out


class Tracking_MPC(nn.Module):
    """Tracks the network's reference with the differentiable MPC (policies.py:567-686)."""

    def __init__(self, args, env):
        super().__init__()
        self.args = args
        self.nu, self.nx, self.nq, self.dt, self.T = env.nu, env.nx, getattr(env, "nq", args.nq), env.dt, args.T
        dyn, dyn_jac = env.dynamics, env.dynamics_derivatives
        if RECOGNISE_ENV_DYNAMICS and not isinstance(dyn, DeviceDynamics):
            # the reference passes its env's torch module (policies.py:571-572): if it IS one of the registered device
            # models (checked numerically on the module itself, dynamics.recognise), the solver calls stay on chip
            known = recognise(dyn, env.nx, env.nu, dt=getattr(env, "dt", None), device=args.device)
            if known is not None:
                dyn, dyn_jac = known, known.jac
        self.dyn, self.dyn_jac = dyn, dyn_jac
        self.device = args.device
        self.dtype = torch.float64 if args.dtype == "double" else torch.float32
        self.u_upper = torch.as_tensor(env.action_space.high).to(self.device)
        self.u_lower = torch.as_tensor(env.action_space.low).to(self.device)
        self.qp_iter, self.eps, self.warm_start, self.bsz = args.qp_iter, args.eps, args.warm_start, args.bsz
        if args.Q is None:
            Q = torch.ones(self.nx, dtype=self.dtype, device=self.device)
            R = torch.ones(self.nu, dtype=self.dtype, device=self.device)
        else:
            Q, R = args.Q.to(self.device), args.R.to(self.device)
        Qd = torch.cat([Q, R], dim=0).to(self.dtype)
        self.Q = torch.diag(Qd).repeat(self.bsz, self.T, 1, 1)
        self.u_init = torch.randn(self.bsz, self.T, self.nu, dtype=self.dtype, device=self.device)
        self.x_init = None
        self.single_qp_solve = self.qp_iter == 1
        if args.solver_type == "al":
            self.ctrl = al_mpc.MPC(self.nx, self.nu, self.T, u_lower=self.u_lower, u_upper=self.u_upper,
                                   exit_unconverged=False, eps=1e-5, n_batch=self.bsz, backprop=False, verbose=0,
                                   u_init=self.u_init, solver_type="dense", dtype=self.dtype)
        else:
            self.ctrl = ip_mpc.MPC(self.nx, self.nu, self.T, u_lower=self.u_lower.double(), u_upper=self.u_upper.double(),
                                   qp_iter=self.qp_iter, exit_unconverged=False, eps=1e-5, n_batch=self.bsz,
                                   backprop=False, verbose=0, u_init=self.u_init.transpose(0, 1),
                                   grad_method=ip_mpc.GradMethods.ANALYTIC, solver_type="dense",
                                   single_qp_solve=self.single_qp_solve)

    def compute_p(self, x_ref):
        self.p = -(self.Q * x_ref.unsqueeze(-2)).sum(dim=-1)                      # policies.py:669-680
        return self.p

    def forward(self, x0, xu_ref, x_ref, u_ref):
        al = self.args.solver_type == "al"
        if al:
            xu_ref = torch.cat([x_ref, u_ref], dim=-1)
            if self.x_init is None:
                self.x_init = self.ctrl.x_init = x_ref
                self.u_init = self.ctrl.u_init = u_ref
        self.compute_p(xu_ref)
        if al:
            cost = al_utils.QuadCost(self.Q, self.p)
            xs, us = self.ctrl(x0, cost, self.dyn, self.dyn_jac)
        else:
            cost = ip_mpc.QuadCost(self.Q.transpose(0, 1), self.p.transpose(0, 1))
            self.ctrl.u_init = self.u_init.transpose(0, 1)
            xs, us = self.ctrl(x0.to(self.dtype), cost, self.dyn, self.dyn_jac)
            xs, us = xs.transpose(0, 1).to(x0.dtype), us.transpose(0, 1).to(x0.dtype)     # the network's dtype
        self.u_init = us.clone().detach()
        return xs, us

    def reinitialize(self, x, mask):
        self.u_init = torch.randn(self.bsz, self.T, self.nu, dtype=x.dtype, device=x.device)
        self.x_init = None
        self.ctrl.reinitialize(x, mask)


class DEQMPCPolicy(nn.Module):
    """deq_iter rounds of [network proposes a reference -> MPC tracks it -> the solution feeds the
    network again] (policies.py:432-529).  Returns every round's (network reference, MPC states,
    MPC actions): the imitation loss supervises all of them."""

    def __init__(self, args, env):
        super().__init__()
        self.args = args
        self.nu, self.nx, self.nq, self.T, self.dt = env.nu, env.nx, args.nq, args.T, env.dt
        self.device, self.deq_iter = args.device, args.deq_iter
        self.model = DEQLayer(args, env).to(self.device)
        self.out_type = args.policy_out_type
        self.tracking_mpc = Tracking_MPC(args, env)

    def forward(self, x, x_gt, u_gt, mask, iter=0, qp_solve=True, lastqp_solve=False):
        bsz = x.shape[0]
        x_ref = torch.cat([x] * self.T, dim=-1).detach().clone()
        nominal_actions = torch.zeros((bsz, self.T, self.nu), device=self.device)
        z = self.model.init_z(bsz)
        trajs = []
        if self.args.solver_type == "al":
            self.tracking_mpc.reinitialize(x, mask[:, :, None])
        for _ in range(self.deq_iter):
            x_ref, z = self.model(x_ref, z)
            if self.model.out_type == 1:                      # the current state is known: prepend it
                x_ref = torch.cat([x[:, None, :], x_ref.view(-1, self.T - 1, self.nx)], dim=1)
            else:
                x_ref = x_ref.view(-1, self.T, self.nx)
            xu_ref = torch.cat([x_ref, nominal_actions], dim=-1)
            x_ref_tr, u_ref_tr = x_ref, nominal_actions
            nominal_states = x_ref
            if qp_solve:
                nominal_states, nominal_actions = self.tracking_mpc(x, xu_ref, x_ref_tr, u_ref_tr)
            trajs.append((x_ref, nominal_states, nominal_actions))
            x_ref = nominal_states.reshape(bsz, -1).detach().clone()              # the solution feeds the DEQ again
        with torch.no_grad():
            nxt = self.tracking_mpc.dyn(x_ref.view(-1, self.nx).double(), u_gt.reshape(-1, self.nu).double())
            dyn_res = nxt.reshape(bsz, -1).norm(dim=1).mean()
        if lastqp_solve:                                      # policies.py:524-527
            ns, na = self.tracking_mpc(x, xu_ref, x_ref_tr, u_ref_tr)
            trajs[-1] = (trajs[-1][0], ns, na)
        return trajs, dyn_res


class FFDNetwork(nn.Module):
    """Feed-forward reference generator of NNMPCPolicy (policies.py:532-564): state -> T configurations, as offsets
    from the current configuration.  Layer names as the reference's (fc1 .. fc3 / `net`) for its state dicts."""

    def __init__(self, args, env):
        super().__init__()
        self.args = args
        self.nu, self.nx, self.nq, self.T = env.nu, env.nx, args.nq, args.T
        self.fc1, self.ln1, self.relu1 = nn.Linear(self.nx, 256), nn.LayerNorm(256), nn.ReLU()
        self.fc2, self.ln2, self.relu2 = nn.Linear(256, 256), nn.LayerNorm(256), nn.ReLU()
        self.fc3 = nn.Linear(256, self.nq * self.T)
        self.net = nn.Sequential(self.fc1, self.ln1, self.relu1, self.fc2, self.ln2, self.relu2, self.fc3)

    def forward(self, x):
        return self.net(x).view(-1, self.T, self.nq) + x[:, None, :self.nq]


class NNMPCPolicy(nn.Module):
    """Feed-forward network proposes a configuration trajectory, the differentiable MPC tracks it
    (policies.py:689-716).  The reference's forward hands Tracking_MPC two arguments in a time-major layout its
    four-argument, batch-major forward (policies.py:640) no longer takes; here the same reference -- configurations
    from the network, zero velocities, zero controls -- goes through the current signature.  Call `reinitialize` (or
    pass `mask`) before the first AL solve, as AL_mpc.MPC requires (AL_mpc.py:432)."""

    def __init__(self, args, env):
        super().__init__()
        self.args = args
        self.nu, self.nx, self.nq, self.T, self.dt = env.nu, env.nx, args.nq, args.T, env.dt
        self.device = args.device
        self.out_type = args.policy_out_type
        self.model = FFDNetwork(args, env).to(self.device)
        self.tracking_mpc = Tracking_MPC(args, env)

    def reinitialize(self, x, mask):
        self.tracking_mpc.reinitialize(x, mask)

    def forward(self, x, mask=None):
        q_ref = self.model(x)
        x_ref = torch.cat([q_ref, torch.zeros(q_ref.shape[:-1] + (self.nx - self.nq,), dtype=q_ref.dtype, device=q_ref.device)], dim=-1)
        u_ref = torch.zeros(x.shape[0], self.T, self.nu, dtype=q_ref.dtype, device=q_ref.device)
        if self.args.solver_type == "al":
            if mask is None:
                mask = torch.ones(x.shape[0], self.T, 1, dtype=x.dtype, device=x.device)
            self.tracking_mpc.reinitialize(x, mask)
        return self.tracking_mpc(x, torch.cat([x_ref, u_ref], dim=-1), x_ref, u_ref)


class NNPolicy(nn.Module):
    """Behaviour-cloning baseline without a solver (policies.py:719-784): state -> trajectory of actions (out_type 0),
    states (1), both (2) or configurations with finite-difference velocities (3)."""

    def __init__(self, args, env):
        super().__init__()
        self.args = args
        self.nu, self.nx, self.nq, self.T, self.dt = env.nu, env.nx, args.nq, args.T, env.dt
        self.device, self.hdim, self.out_type = args.device, args.hdim, args.policy_out_type
        self.out_dim = {0: self.nu, 1: self.nx, 2: self.nx + self.nu, 3: self.nq}[self.out_type] * self.T
        self.model = nn.Sequential(nn.Linear(self.nx, self.hdim), nn.LayerNorm(self.hdim), nn.ReLU(),
                                   nn.Linear(self.hdim, self.hdim), nn.LayerNorm(self.hdim), nn.ReLU())
        self.model.add_module("out", nn.Linear(self.hdim, self.out_dim))

    def forward(self, x):
        y = self.model(x)
        if self.out_type == 0:
            return None, y.view(-1, self.T, self.nu)
        if self.out_type == 1:
            return y.view(-1, self.T, self.nx), None
        if self.out_type == 2:
            cut = self.nx * self.T
            return y[:, :cut].view(-1, self.T, self.nx), y[:, cut:].view(-1, self.T, self.nu)
        # out_type 3.  The reference differences the FLAT (bsz, nq T) output along its last axis' neighbour
        # (`pos[:, 1:] - pos[:, :-1]`, policies.py:779-781), which mixes coordinates for nq > 1 and leaves shapes that
        # only concatenate for nq = 1; here the difference is taken per knot, which is the same thing at nq = 1.
        pos = y.view(-1, self.T, self.nq)
        vel = (pos[:, 1:] - pos[:, :-1]) / self.dt
        vel = torch.cat([vel, vel[:, -1:]], dim=1)
        return torch.cat([pos, vel], dim=-1), None


def add_loss_based_on_out_type(policy, out_type, gt_states, gt_actions, gt_mask, nominal_states, nominal_actions):
    """policies.py:818-833: masked L1 on actions (0, 2), states (1, 2) or configurations (3)."""
    loss = 0.0
    m = gt_mask[:, :, None]
    if out_type in (0, 2):
        loss = loss + torch.abs((nominal_actions - gt_actions) * m).sum(dim=-1).mean()
    if out_type in (1, 2):
        loss = loss + torch.abs((nominal_states - gt_states) * m).sum(dim=-1).mean()
    if out_type == 3:
        loss = loss + torch.abs((nominal_states[..., :policy.nq] - gt_states[..., :policy.nq]) * m).sum(dim=-1).mean()
    return loss


def compute_loss_deqmpc(policy, gt_states, gt_actions, gt_mask, trajs):
    """policies.py:800-808: every DEQ-MPC round is supervised; loss_end is the last round's term."""
    loss = 0.0
    for (_, ns, na) in trajs:
        loss = loss + add_loss_based_on_out_type(policy, policy.out_type, gt_states, gt_actions, gt_mask, ns, na)
    _, ns, na = trajs[-1]
    return loss, add_loss_based_on_out_type(policy, policy.out_type, gt_states, gt_actions, gt_mask, ns, na)


def compute_loss_deq(policy, gt_states, gt_actions, gt_mask, trajs):
    """policies.py:787-797 (DEQ pre-training, `policy` = the DEQLayer): every round's state prediction, out_type 1."""
    loss = 0.0
    for (_, ns, na) in trajs:
        loss = loss + add_loss_based_on_out_type(policy, 1, gt_states, gt_actions, gt_mask, ns, na)
    _, ns, na = trajs[-1]
    return loss, add_loss_based_on_out_type(policy, 1, gt_states, gt_actions, gt_mask, ns, na)


def compute_loss_bc(policy, gt_states, gt_actions, gt_mask, trajs):
    """policies.py:811-816: one (states, actions) pair from NNPolicy / NNMPCPolicy."""
    ns, na = trajs
    return add_loss_based_on_out_type(policy, policy.out_type, gt_states, gt_actions, gt_mask, ns, na), torch.zeros(1)


def compute_loss(policy, gt_states, gt_actions, gt_mask, trajs, args):
    """policies.py:836-848: the loss of the training mode `args` selects."""
    if args.deq:
        if args.en_qp_solve:
            return compute_loss_deqmpc(policy, gt_states, gt_actions, gt_mask, trajs)
        return compute_loss_deq(policy.model, gt_states, gt_actions, gt_mask, trajs)
    return compute_loss_bc(policy, gt_states, gt_actions, gt_mask, trajs)


def allreduce_gradients(module, group=None, world_size=None):
    """Sum the gradients of `module` over the ranks of `group` with ONE flat all_reduce (the DEQLayer
    has ~0.1-1 MB of parameters: a single latency-bound collective over xGMI beats per-tensor calls),
    then divide by the world size: every rank's loss is a mean over ITS shard, so the average of the
    rank gradients is the gradient of the mean over the global batch for equal shards."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    ws = world_size or dist.get_world_size(group)
    if ws == 1:
        return 0
    ps = [p for p in module.parameters() if p.requires_grad]
    for p in ps:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    flat = torch.cat([p.grad.reshape(-1) for p in ps])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(ws)
    off = 0
    for p in ps:
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n
    return flat.numel()


def train_step(policy, optimizer, x0, gt_states, gt_actions, gt_mask, group=None, qp_solve=True,
               lastqp_solve=False):
    """One imitation-learning step on this rank's shard (deqmpc/train.py:150-175): forward through
    deq_iter solver calls, L1 loss on every iterate, backward through the solvers' implicit
    derivatives, gradient all-reduce, optimiser step.  Returns (loss, loss_end, dyn_res) tensors."""
    trajs, dyn_res = policy(x0, gt_states, gt_actions, gt_mask, qp_solve=qp_solve, lastqp_solve=lastqp_solve)
    loss, loss_end = compute_loss_deqmpc(policy, gt_states, gt_actions, gt_mask, trajs)
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    allreduce_gradients(policy.model, group)
    optimizer.step()
    return loss.detach(), loss_end.detach(), dyn_res


class GraphedTrainStep:
    """train_step captured as ONE hipGraph: deq_iter x [DEQLayer -> AL_mpc.MPC solve], the loss, the backward through
    the solvers and -- with a capturable optimiser and no process group -- the optimiser step.  Every solver call is
    a C-ABI call that only enqueues on the current stream (dqp_al_mpc_solve, dqp_al_banded_solve), so a training step
    becomes one graph launch instead of several hundred Python-dispatched ones: this is the small-batch path (the
    reference trains at --bsz 128, deqmpc/train.py:46, where a step is launch-bound).  Shapes and the batch size are
    frozen at capture; `__call__` copies the batch into the graph's static inputs.  With a process group the graph
    ends after backward and the flat gradient all-reduce and the optimiser step run eagerly after the replay.
    The Cholesky-failure flags cannot be looked at inside a graph (AL_mpc.CHECK_CHOLESKY is off for the captured
    calls): `failed()` reads the flags of the last replay (one synchronisation) -- a set flag means that step's
    solve should be redone eagerly."""

    def __init__(self, policy, optimizer, x0, gt_states, gt_actions, gt_mask, group=None, warmup=3):
        self.policy, self.opt, self.group = policy, optimizer, group
        self.static = [t.detach().clone() for t in (x0, gt_states, gt_actions, gt_mask)]
        self.step_in_graph = group is None and all(g.get("capturable", False) for g in optimizer.param_groups)
        self._flags = []
        check = al_mpc.CHECK_CHOLESKY
        al_mpc.CHECK_CHOLESKY = False
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    optimizer.zero_grad(set_to_none=True)
                    self._body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            optimizer.zero_grad(set_to_none=True)
            with torch.cuda.graph(self.graph):
                self.out = self._body()
                self._flags = list(getattr(policy.tracking_mpc.ctrl, "fail_log", []))
        finally:
            al_mpc.CHECK_CHOLESKY = check

    def _body(self):
        ctrl = self.policy.tracking_mpc.ctrl
        if hasattr(ctrl, "fail_log"):
            ctrl.fail_log = []
        trajs, dyn_res = self.policy(*self.static)
        loss, loss_end = compute_loss_deqmpc(self.policy, self.static[1], self.static[2], self.static[3], trajs)
        loss.backward()
        if self.step_in_graph:
            self.opt.step()
        return loss.detach(), loss_end.detach(), dyn_res

    def __call__(self, x0, gt_states, gt_actions, gt_mask):
        for dst, src in zip(self.static, (x0, gt_states, gt_actions, gt_mask)):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        self.graph.replay()
        if not self.step_in_graph:
            allreduce_gradients(self.policy.model, self.group)
            self.opt.step()
        return self.out

    def failed(self):
        return any(bool(f.any()) for f in self._flags)
