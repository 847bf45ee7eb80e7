"""Batch sharding of independent QPs across ranks.

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the
CPU tests).  QPs are independent, so every batched input is split into contiguous B/world
slices, parameters without a batch dim are replicated, and the only data-path collective is one
all_gather of the solved shard.  Gradients of shared parameters follow the reference's
`.mean(0)` over the FULL batch (qpth/qp.py:160-178): all_reduce(sum of local per-sample
grads) / B.
"""
import torch
import torch.distributed as dist


def shard_bounds(nbatch, world, rank):
    """Contiguous, balanced partition: the first (nbatch % world) ranks get one extra QP."""
    base, rem = divmod(nbatch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_params(params, dims, world, rank):
    """params: (Q,p,G,h,A,b); dims: their batched ndim (3,2,3,2,3,2).  Returns local views."""
    nbatch = 1
    for t, nd in zip(params, dims):
        if t.numel() > 0 and t.dim() == nd:
            nbatch = t.shape[0]
            break
    lo, hi = shard_bounds(nbatch, world, rank)
    out = []
    for t, nd in zip(params, dims):
        out.append(t[lo:hi] if (t.numel() > 0 and t.dim() == nd) else t)
    return out, (lo, hi, nbatch)


def gather_solution(z_local, nbatch, group=None):
    """The single collective of the forward path: all_gather of the (B_local, nz) shard.
    Handles ragged shards (nbatch % world != 0) by padding to the largest shard."""
    world = dist.get_world_size(group)
    sizes = [shard_bounds(nbatch, world, r) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = z_local
    if z_local.shape[0] < mx:
        pad = torch.zeros(mx, *z_local.shape[1:], dtype=z_local.dtype, device=z_local.device)
        pad[: z_local.shape[0]] = z_local
    buf = torch.empty(world * mx, *z_local.shape[1:], dtype=z_local.dtype, device=z_local.device)
    dist.all_gather_into_tensor(buf, pad.contiguous(), group=group)
    parts = [buf[r * mx: r * mx + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(parts, 0)


class _GatherWithGrad(torch.autograd.Function):
    """gather_solution with a backward: a loss taken on the gathered batch sends every rank the
    cotangent slice of its own shard (each rank evaluates the same loss on the same gathered
    tensor, so no reduction is needed -- the slices of other ranks belong to their own graphs)."""

    @staticmethod
    def forward(ctx, z_local, nbatch, group):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        ctx.bounds = shard_bounds(nbatch, world, rank)
        return gather_solution(z_local.detach(), nbatch, group)

    @staticmethod
    def backward(ctx, g):
        lo, hi = ctx.bounds
        return g[lo:hi].contiguous(), None, None


def gather_solution_autograd(z_local, nbatch, group=None):
    """Differentiable form of gather_solution (use when the loss is a function of the full batch)."""
    return _GatherWithGrad.apply(z_local, nbatch, group)


class global_batch_rule:
    """Context manager: inside it QPFunction / DenseQPFunction evaluate the reference's batch-coupled stop
    (batch.py:119-144) over the WHOLE sharded batch instead of this rank's shard -- one extra collective,
    a bitwise-OR all_reduce of three int64 iteration masks (24 bytes), between pass 1 and the decision
    (include/dqp.h: dqp_term_local_masks / dqp_qp_forward_finish).  Results on the shards are then bit-identical
    to a single-device solve of the concatenated batch.

        with sharding.global_batch_rule(group):
            z_local = QPFunction()(Q[lo:hi], p[lo:hi], G[lo:hi], h[lo:hi], A[lo:hi], b[lo:hi])
    """

    def __init__(self, group=None):
        self.group = group

    def _exchange(self, masks):
        if dist.get_backend(self.group) == "gloo" and masks.is_cuda:      # (CPU rehearsal of the RCCL path)
            host = masks.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.BOR, group=self.group)
            return host.to(masks.device)
        dist.all_reduce(masks, op=dist.ReduceOp.BOR, group=self.group)
        return masks

    def __enter__(self):
        from . import qp
        self._old, qp.MASK_EXCHANGE = qp.MASK_EXCHANGE, self._exchange
        return self

    def __exit__(self, *exc):
        from . import qp
        qp.MASK_EXCHANGE = self._old
        return False


def reduce_shared_grad_from_local_mean(g_local_mean, n_local, nbatch, group=None):
    """QPFunction.backward hands back the LOCAL `.mean(0)` for a parameter shared by the batch
    (qp.py:160-178 semantics on this rank's shard).  The reference's value on the full batch is the
    mean over all nbatch samples: weight every rank's mean by its shard size (ragged shards
    included), sum, divide by nbatch."""
    return reduce_shared_grad(g_local_mean * float(n_local), nbatch, group)


def reduce_shared_grad(g_local_sum, nbatch, group=None):
    """g_local_sum: SUM over the local shard of per-sample gradients of a shared parameter."""
    g = g_local_sum.clone()
    dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
    return g / nbatch


def solve_sharded(solve_fn, params, dims=(3, 2, 3, 2, 3, 2), group=None):
    """Shard -> local solve (solve_fn(*local_params) -> (B_local, nz)) -> one all_gather."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    local, (lo, hi, nbatch) = shard_params(params, dims, world, rank)
    z_local = solve_fn(*local)
    if z_local.requires_grad:
        return gather_solution_autograd(z_local, nbatch, group), (lo, hi)
    return gather_solution(z_local, nbatch, group), (lo, hi)
