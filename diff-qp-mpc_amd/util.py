"""Batch-shape helpers of the operator mirrors.

They reproduce the *behaviour* of the reference's helpers of the same names (qpth/util.py:22-51,
88-89: a parameter that has one dimension fewer than its batched form is shared by every problem
of the batch), because callers of qpth.qp.QPFunction rely on those broadcasting rules; the
implementations are this package's own.
"""
import torch

# batched rank of each QP parameter, in call order (Q, p, G, h, A, b)
_BATCHED_RANK = {"Q": 3, "p": 2, "G": 3, "h": 2, "A": 3, "b": 2}


def extract_nBatch(Q, p, G, h, A, b):
    """Batch size = leading extent of the first parameter given in batched form, else 1."""
    for t, rank in zip((Q, p, G, h, A, b), _BATCHED_RANK.values()):
        if t.dim() == rank:
            return t.shape[0]
    return 1


def expandParam(X, nBatch, nDim):
    """-> (tensor viewed with a batch axis, was_shared).  Scalars, empty tensors and tensors that
    already have `nDim` axes pass through; one axis fewer means shared across the batch (a
    stride-0 view, no copy); anything else is an error, as in the reference."""
    rank = X.dim()
    if rank == nDim or rank == 0 or X.numel() == 0:
        return X, False
    if rank + 1 != nDim:
        raise RuntimeError("Unexpected number of dimensions.")
    return X[None].expand(nBatch, *X.shape), True


def get_sizes(G, A=None):
    """-> (nineq, nz, neq, nBatch) from G (nineq,nz) or (B,nineq,nz) and an optional A; neq is None
    without A and 0 for an empty A.  (neq is read from axis 1 of A like the reference does, which is
    only meaningful for a batched A.)"""
    if G.dim() not in (2, 3):
        raise RuntimeError("G must be (nineq, nz) or (B, nineq, nz)")
    nBatch = G.shape[0] if G.dim() == 3 else 1
    nineq, nz = G.shape[-2], G.shape[-1]
    neq = None
    if A is not None:
        neq = A.shape[1] if A.numel() > 0 else 0
    return nineq, nz, neq, nBatch


def bger(x, y):
    """Batched outer product: (B,n), (B,m) -> (B,n,m)."""
    return torch.einsum("bi,bj->bij", x, y)
