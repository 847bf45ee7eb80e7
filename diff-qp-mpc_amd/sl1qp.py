"""The l1-slack ("SL1QP") reformulation of a QP the reference sketches in qpth/sl1qp_mpc.py:703-752
(`sl1qpify`), implemented for general sizes.

    original      min_z 1/2 z'Qz + p'z            s.t.  Gz <= h,  Az = b
    reformulated  min   1/2 z'Qz + p'z + mu 1'(v + w) + mu 1't
                  s.t.  Gz - h <= t,   Az - b = v - w,   v, w, t >= 0

with variables [z (nz); v (neq); w (neq); t (nineq)].  The reference's code only type-checks when
neq == nineq (it sizes the equality slacks with the inequality count and vice versa,
sl1qp_mpc.py:723-751), its MPC clone stops at an unconditional ipdb.set_trace() (:326), and its Q block
for the slacks is exactly zero, which qpth's own `Q is not SPD` check rejects.  Here the blocks have the
right shapes and the slack block of Q is `reg` I (default 1e-6: the PDIPM factorises Q by Cholesky);
for mu above the largest multiplier of the original QP the penalty is exact, so z is the original
solution up to O(reg).

    z = SL1QPFunction(mu=50.0)(Q, p, G, h, A, b)          # (B, nz), differentiable wrt all six
"""
import torch

from .qp import DenseQPFunction
from .util import expandParam, extract_nBatch


def sl1qpify(Q, p, G, h, A, b, mu, reg=1e-6):
    """(B,..)-batched dense QP -> the reformulated (Q', p', G', h', A', b'); sizes
    nz' = nz + 2 neq + nineq, nineq' = 2 nineq + 2 neq, neq' = neq."""
    B = extract_nBatch(Q, p, G, h, A, b)
    Q, _ = expandParam(Q, B, 3); p, _ = expandParam(p, B, 2)
    G, _ = expandParam(G, B, 3); h, _ = expandParam(h, B, 2)
    A, _ = expandParam(A, B, 3); b, _ = expandParam(b, B, 2)
    nz, nineq, neq = Q.shape[-1], G.shape[-2], A.shape[-2]
    kw = dict(dtype=Q.dtype, device=Q.device)
    zeros = lambda r, c: torch.zeros(B, r, c, **kw)
    eye = lambda n: torch.eye(n, **kw).expand(B, n, n)
    ns = 2 * neq + nineq
    Q2 = torch.cat((torch.cat((Q, zeros(nz, ns)), 2),
                    torch.cat((zeros(ns, nz), reg * eye(ns)), 2)), 1)
    p2 = torch.cat((p, torch.full((B, ns), float(mu), **kw)), 1)
    A2 = torch.cat((A, -eye(neq), eye(neq), zeros(neq, nineq)), 2)                 # Az - v + w = b
    G2 = torch.cat((torch.cat((G, zeros(nineq, 2 * neq), -eye(nineq)), 2),           # Gz - t <= h
                    torch.cat((zeros(ns, nz), -eye(ns)), 2)), 1)                     # -(v, w, t) <= 0
    h2 = torch.cat((h, torch.zeros(B, ns, **kw)), 1)
    return Q2, p2, G2, h2, A2, b


def SL1QPFunction(mu=10.0, reg=1e-6, **solver_kw):
    """Callable (Q, p, G, h, A, b) -> z of the l1-penalised problem, solved by DenseQPFunction on the MI355X kernels:
    the one-wavefront kernels up to 64 extended variables, the blocked dense kernels (csrc/dqp_big.hip) above -- MPC
    shapes: 90 extended variables at n 3 m 3 T 5, 120 at the pendulum's T 10 (limit: 512, include/dqp.h)."""
    solve = DenseQPFunction(**solver_kw)

    def apply(Q, p, G, h, A, b):
        nz = Q.shape[-1]
        ext = sl1qpify(Q, p, G, h, A, b, mu, reg)
        return solve(*ext)[..., :nz]
    return apply
