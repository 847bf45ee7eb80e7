"""Broadcasting helpers with the reference's semantics (qpth/util.py:22-51, 88-89)."""
import torch


def get_sizes(G, A=None):
    """qpth/util.py:22-33"""
    if G.dim() == 2:
        nineq, nz = G.size()
        nBatch = 1
    elif G.dim() == 3:
        nBatch, nineq, nz = G.size()
    if A is not None:
        neq = A.size(1) if A.nelement() > 0 else 0
    else:
        neq = None
    return nineq, nz, neq, nBatch


def expandParam(X, nBatch, nDim):
    """qpth/util.py:36-43: a parameter with one fewer dim is shared by the whole batch."""
    if X.ndimension() in (0, nDim) or X.nelement() == 0:
        return X, False
    elif X.ndimension() == nDim - 1:
        return X.unsqueeze(0).expand(*([nBatch] + list(X.size()))), True
    else:
        raise RuntimeError("Unexpected number of dimensions.")


def extract_nBatch(Q, p, G, h, A, b):
    """qpth/util.py:46-51"""
    dims = [3, 2, 3, 2, 3, 2]
    params = [Q, p, G, h, A, b]
    for param, dim in zip(params, dims):
        if param.ndimension() == dim:
            return param.size(0)
    return 1


def bger(x, y):
    """qpth/util.py:88-89"""
    return x.unsqueeze(2).bmm(y.unsqueeze(1))
