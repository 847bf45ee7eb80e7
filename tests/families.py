"""Seeded problem families shared by the tests, tools/stress_parity.py and the golden generators
(SURVEY.md §8d).  numpy fp64 out, batch-major: [Q, p, G, h, A, b]."""
import numpy as np
import torch


def family(seed, B, nz, nineq, neq, kind="R"):
    """kind R: random dense QP of the reference's profilers (prof-linear.py:64-75, test.py:42-55):
    Q = L L^T + 1e-3 I;  kind D: diagonal, well-conditioned (MPC-like) cost.  h = G z0 + s0 and
    b = A z0 make every problem feasible."""
    g = torch.Generator().manual_seed(seed)
    if kind == "R":
        L = torch.randn(B, nz, nz, generator=g, dtype=torch.float64)
        Q = L @ L.transpose(1, 2) + 1e-3 * torch.eye(nz, dtype=torch.float64)
    else:
        Q = torch.diag_embed(torch.rand(B, nz, generator=g, dtype=torch.float64) + 0.1)
    G = torch.randn(B, nineq, nz, generator=g, dtype=torch.float64)
    z0 = torch.randn(B, nz, generator=g, dtype=torch.float64)
    s0 = torch.rand(B, nineq, generator=g, dtype=torch.float64)
    A = torch.randn(B, neq, nz, generator=g, dtype=torch.float64)
    p = torch.randn(B, nz, generator=g, dtype=torch.float64)
    h = (G @ z0.unsqueeze(-1)).squeeze(-1) + s0
    b = (A @ z0.unsqueeze(-1)).squeeze(-1)
    return [t.numpy() for t in (Q, p, G, h, A, b)]


def family_mpc(seed, B, n=3, m=3, T=5):
    """MPC-structured dense QP (family M): block-diagonal cost, dynamics equalities
    x_{t+1} = A x_t + B u_t, x_0 given, box |u| <= 1 (so many constraints are active) -- the
    structure qp_wrapper.compute_*_dense (qp_wrapper.py:638-679) produces."""
    rng = np.random.default_rng(seed)
    nt, nz, neq, nineq = n + m, T * (n + m), T * n, 2 * T * m
    Q = np.tile(np.eye(nz), (B, 1, 1)) * (0.5 + rng.random((B, 1, 1)))
    p = rng.standard_normal((B, nz))
    A = np.zeros((B, neq, nz)); b = np.zeros((B, neq))
    Ad = np.eye(n) + 0.2 * rng.standard_normal((B, n, n)); Bd = rng.standard_normal((B, n, m))
    for t in range(T - 1):
        r0 = t * n
        A[:, r0:r0 + n, t * nt:t * nt + n] = -Ad
        A[:, r0:r0 + n, t * nt + n:(t + 1) * nt] = -Bd
        A[:, r0:r0 + n, (t + 1) * nt:(t + 1) * nt + n] = np.eye(n)
    A[:, (T - 1) * n:, :n] = np.eye(n)
    b[:, (T - 1) * n:] = rng.standard_normal((B, n))
    G = np.zeros((B, nineq, nz)); h = np.ones((B, nineq))
    for t in range(T):
        for i in range(m):
            G[:, t * m + i, t * nt + n + i] = 1.0
            G[:, T * m + t * m + i, t * nt + n + i] = -1.0
    return [np.ascontiguousarray(a) for a in (Q, p, G, h, A, b)]


def broke_down(resid_hist, iters):
    """Samples on which the reference's unguarded get_step (batch.py:211-214) divided by an
    exactly-zero step component: their residual history turns NaN while the batch is still
    iterating (the oracle writes NaN only there or after the batch stopped)."""
    h = resid_hist[:, :iters]
    return np.isnan(h).any(1) & np.isfinite(h[:, 0])
