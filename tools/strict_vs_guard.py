#!/usr/bin/env python3
"""How far is the literal reference (batch.py:211-214: get_step divides by the step, no guard) from the guarded one
(batch_LU.py:203-214, what the kernels implement by default), and where does the GPU stand against each?

Per family (R: random dense, D: diagonal cost, M: MPC-structured; metric shape 30/30/15), SEEDS batches of BATCH:
  * `nan`      samples whose residual history turns NaN in the literal oracle while the batch iterates (a zero step
               component was met: the iterate is frozen);
  * `differ`   samples where literal and guarded oracle outputs (zhat, lam, nu, slack) or gradients (all six, random
               cotangent, where strict complementarity holds) differ beyond the test tolerances -- the only samples on
               which "which reference?" matters;
  * GPU default (guarded): worst deviation / tolerance against the LITERAL oracle on the non-differing samples, and
               against the guarded oracle on the differing ones; number of samples over tolerance against the literal one;
  * GPU DQP_FLAG_STRICT_GET_STEP: samples over tolerance against the literal oracle (the flag freezes a problem on an exact
               zero of ITS arithmetic; the reference's zeros are accidents of the reference's arithmetic order).
Writes a JSON summary to stdout's last line (profiles/r3/strict_vs_guard.json keeps a copy)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from oracle import oracle
from diff_qp_mpc_amd import qp as qpmod, _lib
from families import family, family_mpc, broke_down

TOL = {"zhat": (1e-6, 1e-8), "lam": (1e-5, 1e-7), "nu": (1e-5, 1e-7), "slack": (1e-5, 1e-7)}


def over(x, ref):
    """per-sample max |x - ref| / (atol + rtol |ref|) over the four outputs"""
    w = np.zeros(x["zhat"].shape[0])
    for k, (rt, at) in TOL.items():
        e = np.abs(x[k] - ref[k]) / (at + rt * np.abs(ref[k]))
        e = np.where(np.isnan(e), np.inf, e)
        w = np.maximum(w, e.reshape(e.shape[0], -1).max(1))
    return w


GT = (1e-4, 1e-6)


def grad_over(a, b):
    w = np.zeros(a["dQ"].shape[0])
    for k in a:
        e = np.abs(a[k] - b[k]) / (GT[1] + GT[0] * np.abs(b[k]))
        e = np.where(np.isnan(e), np.inf, e)
        w = np.maximum(w, e.reshape(e.shape[0], -1).max(1))
    return w


def gpu(ins, flags, ct):
    dv = [torch.tensor(a, device="cuda") for a in ins]
    qpmod.FORCE_FLAGS = flags
    try:
        zhat, lam, nu, slack, info, resid, saved = qpmod._forward_impl(*dv, 1e-12, 20, 3)
        gr = qpmod._backward_impl(saved, zhat, lam, nu, slack, torch.tensor(ct, device="cuda"), (True,) * 6, flags)
    finally:
        qpmod.FORCE_FLAGS = 0
    torch.cuda.synchronize()
    out = {"zhat": zhat.cpu().numpy(), "lam": lam.cpu().numpy(), "nu": nu.cpu().numpy(), "slack": slack.cpu().numpy()}
    return out, {"d" + k: t.cpu().numpy() for k, t in zip("QpGhAb", gr)}


def main():
    seeds, B = int(os.environ.get("SEEDS", "4")), int(os.environ.get("BATCH", "2048"))
    nz, nineq, neq = 30, 30, 15
    out = {}
    for kind in os.environ.get("KINDS", "R,D,M").split(","):
        acc = dict(samples=0, nan=0, differ=0, conv=0, gpu_over_literal_on_agree=0, gpu_worst_literal_on_agree=0.0,
                   gpu_worst_guard_on_differ=0.0, gpu_over_guard_on_differ=0, strictflag_over_literal=0,
                   strictflag_over_literal_on_differ=0, strictflag_worst_on_agree=0.0)
        for seed in range(seeds):
            ins = family_mpc(seed, B) if kind == "M" else family(1000 * seed + nz, B, nz, nineq, neq, kind)
            lit = oracle.qp_forward(*ins)
            grd = oracle.qp_forward(*ins, guard=True)
            conv = grd["best_resid"] < 1e-8                 # as the tests: problems the oracle itself converged on
            nanm = broke_down(lit["resid_hist"], lit["iters"])
            ct = np.random.default_rng(seed).standard_normal((B, nz))
            bw = lambda o: oracle.qp_backward(ins[0], ins[2], ins[4], o["zhat"], o["lam"], o["nu"], o["slack"], ct)
            g_lit, g_grd = bw(lit), bw(grd)
            sc = np.maximum(grd["lam"], grd["slack"]).min(1) > 1e-5           # strict complementarity (gradients defined)
            differ = ((over(lit, grd) > 1.0) | ((grad_over(g_lit, g_grd) > 1.0) & sc)) & conv
            agree = ~differ & conv
            g0, gg0 = gpu(ins, 0, ct)
            gs, ggs = gpu(ins, _lib.DQP_FLAG_STRICT_GET_STEP, ct)
            o_lit = np.maximum(over(g0, lit), np.where(sc, grad_over(gg0, g_lit), 0.0))
            o_grd = np.maximum(over(g0, grd), np.where(sc, grad_over(gg0, g_grd), 0.0))
            s_lit = np.maximum(over(gs, lit), np.where(sc, grad_over(ggs, g_lit), 0.0))
            acc["samples"] += B; acc["conv"] += int(conv.sum()); acc["nan"] += int(nanm.sum()); acc["differ"] += int(differ.sum())
            acc["gpu_over_literal_on_agree"] += int((o_lit[agree] > 1.0).sum())
            acc["gpu_worst_literal_on_agree"] = max(acc["gpu_worst_literal_on_agree"], float(o_lit[agree].max(initial=0.0)))
            acc["gpu_over_guard_on_differ"] += int((o_grd[differ] > 1.0).sum())
            acc["gpu_worst_guard_on_differ"] = max(acc["gpu_worst_guard_on_differ"], float(o_grd[differ].max(initial=0.0)))
            acc["strictflag_over_literal"] += int((s_lit[conv] > 1.0).sum())
            acc["strictflag_over_literal_on_differ"] += int((s_lit[differ] > 1.0).sum())
            acc["strictflag_worst_on_agree"] = max(acc["strictflag_worst_on_agree"], float(s_lit[agree].max(initial=0.0)))
            print(kind, "seed", seed, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in acc.items()}, flush=True)
        acc["frac_differ"] = acc["differ"] / max(acc["conv"], 1)
        acc["frac_nan"] = acc["nan"] / acc["samples"]
        out[kind] = acc
    print(json.dumps(out))


if __name__ == "__main__":
    main()
