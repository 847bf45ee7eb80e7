#!/usr/bin/env python3
"""Golden vectors for the samples on which the reference's batch.py get_step breaks down.

batch.py:211-214 computes a = -v/dv with no guard; when one component of a step is exactly 0.0
the sample's alpha is -inf, its iterate NaN from then on, and the returned best iterate is frozen
short of convergence.  The reference's other PDIPM module guards this (batch_LU.py:204-210,
`a[dv == 0] = 1.0`).  This script runs the reference (imported from /root/reference, build
container only) on 8 family-M problems -- among them the three that tools/stress_parity.py found
(seed 0 #174 and #1889, seed 1 #1206) -- twice: as is, and with pdipm_b.get_step replaced by
the reference's own batch_LU.get_step.  Both results are stored; tests pin the strict oracle with
the first and the guarded oracle (oracle.qp_forward(guard=True)) with the second.

Usage:  python tests/golden/make_golden_guard.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("DQP_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.dirname(HERE))
m = types.ModuleType("ipdb")
def _st(*a, **k):
    raise RuntimeError("ipdb.set_trace() reached inside the reference")
m.set_trace = _st
sys.modules["ipdb"] = m
sys.path.insert(0, REF)
torch.set_default_dtype(torch.float64)

from qpth.qp import QPFunction  # noqa: E402
from qpth.solvers.pdipm import batch as pdipm_b  # noqa: E402
from qpth.solvers.pdipm import batch_LU as pdipm_lu  # noqa: E402
from families import family_mpc  # noqa: E402

pick = [(0, 174), (0, 1889), (1, 1206), (0, 0), (0, 1), (1, 2), (1, 3), (0, 305)]
batches = {s: family_mpc(s, 2048) for s in (0, 1)}
ins = [np.stack([batches[s][k][i] for s, i in pick]) for k in range(6)]
B, nz = ins[1].shape
rng = np.random.default_rng(7)
ct = rng.standard_normal((B, nz))
out = {("in_" + k): a for k, a in zip("QpGhAb", ins)}
out["ct"] = ct

for tag, step in (("strict", pdipm_b.get_step), ("guard", pdipm_lu.get_step)):
    pdipm_b.get_step = step
    ts = [torch.tensor(a, requires_grad=True) for a in ins]
    Q, p, G, h, A, b = ts
    fn = QPFunction(check_Q_spd=False, verbose=-1)
    hist = []
    zhat = fn(Q, p, G, h, A, b, lambda x: (A @ x.unsqueeze(-1)).squeeze(-1) - b,
              lambda x: (Q @ x.unsqueeze(-1)).squeeze(-1) + p)
    zhat.backward(torch.tensor(ct))
    out[tag + "_zhat"] = zhat.detach().numpy()
    for k, t in zip("QpGhAb", ts):
        out[tag + "_d" + k] = t.grad.numpy()
    # duals as the reference stashes them on ctx: recompute through pdipm_b.forward directly
    with torch.no_grad():
        Q_LU, S_LU, R = pdipm_b.pre_factor_kkt(Q, G, A)
        x, y, z, s = pdipm_b.forward(Q, p, G, h, A, b, Q_LU, S_LU, R,
                                     lambda x: (A @ x.unsqueeze(-1)).squeeze(-1) - b,
                                     lambda x: (Q @ x.unsqueeze(-1)).squeeze(-1) + p,
                                     1e-12, -1, 3, 20)
    out[tag + "_lam"], out[tag + "_nu"], out[tag + "_slack"] = z.numpy(), y.numpy(), s.numpy()
    assert np.abs(x.numpy() - out[tag + "_zhat"]).max() == 0.0

np.savez_compressed(os.path.join(HERE, "Mz_guard_b8.npz"), **out)
d = np.abs(out["strict_slack"] - out["guard_slack"]).max(1)
print("per-sample max |slack_strict - slack_guard|:", d)
print("per-sample max |dQ_strict - dQ_guard|:", np.abs(out["strict_dQ"] - out["guard_dQ"]).reshape(B, -1).max(1))
