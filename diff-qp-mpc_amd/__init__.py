"""MI355X-native differentiable batched QP solver (drop-in for the qpth hot path of
swami1995/diff-qp-mpc).

Host-side mirror of the reference operator interface; all compute is in hand-written HIP
kernels (csrc/) reached through the C ABI declared in include/dqp.h.  There is NO CPU or
PyTorch fallback: if the HIP library is missing or the tensors are not on a GPU the
operators raise.

    from diff_qp_mpc_amd import QPFunction, DenseQPFunction       # qpth/qp.py:19, :187
    from diff_qp_mpc_amd.qp_wrapper import MPC, QuadCost, LinDx   # qpth/qp_wrapper.py
"""
from .qp import QPFunction, DenseQPFunction, QPSolvers  # noqa: F401
from . import _lib  # noqa: F401

__all__ = ["QPFunction", "DenseQPFunction", "QPSolvers"]
