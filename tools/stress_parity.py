#!/usr/bin/env python3
"""Wide parity sweep on the GPU box: many seeds x the DPP-row sizes x both forward families,
GPU (through the Python mirror -> C ABI) against the CPU oracle, per problem.  Problems the
oracle itself did not converge on (best residual >= 1e-8) are masked, as in the tests.
Prints the worst deviations; exits non-zero if any exceeds the test tolerances."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle
from diff_qp_mpc_amd import qp as qpmod, _lib, _build

def family(seed, B, nz, nineq, neq, kind):
    g = torch.Generator().manual_seed(seed)
    if kind == "R":
        L = torch.randn(B, nz, nz, generator=g, dtype=torch.float64)
        Q = L @ L.transpose(1, 2) + 1e-3 * torch.eye(nz, dtype=torch.float64)
    else:       # well-conditioned, MPC-like cost
        Q = torch.diag_embed(torch.rand(B, nz, generator=g, dtype=torch.float64) + 0.1)
    G = torch.randn(B, nineq, nz, generator=g, dtype=torch.float64)
    z0 = torch.randn(B, nz, generator=g, dtype=torch.float64)
    s0 = torch.rand(B, nineq, generator=g, dtype=torch.float64)
    A = torch.randn(B, neq, nz, generator=g, dtype=torch.float64)
    p = torch.randn(B, nz, generator=g, dtype=torch.float64)
    h = (G @ z0.unsqueeze(-1)).squeeze(-1) + s0
    b = (A @ z0.unsqueeze(-1)).squeeze(-1)
    return [t.numpy() for t in (Q, p, G, h, A, b)]

oracle.build()
qpmod.STALL_TOL = float(os.environ.get("STALL_TOL", qpmod.STALL_TOL))
seeds = int(os.environ.get("SEEDS", "6"))
B = int(os.environ.get("BATCH", "2048"))
worst = {}
bad = 0
t0 = time.time()
for (nz, nineq, neq) in _build.R16N_SIZES:
    for kind in ("R", "D"):
        for seed in range(seeds):
            ins = family(1000 * seed + nz, B, nz, nineq, neq, kind)
            o = oracle.qp_forward(*ins)
            cm = o["best_resid"] < 1e-8
            dv = [torch.tensor(a, device="cuda") for a in ins]
            ct = np.random.default_rng(seed).standard_normal((B, nz))
            og = oracle.qp_backward(ins[0], ins[2], ins[4], o["zhat"], o["lam"], o["nu"], o["slack"], ct)
            gm = cm & (np.maximum(o["lam"], o["slack"]).min(1) > 1e-5)
            for fam, flag in (("nullspace", 0), ("rows", _lib.DQP_FLAG_NO_NULLSPACE)):
                qpmod.FORCE_FLAGS = flag
                zhat, lam, nu, slack, info, resid, saved = qpmod._forward_impl(*dv, float(os.environ.get('EPS', '1e-12')), 20, 3)
                gr = qpmod._backward_impl(saved, zhat, lam, nu, slack, torch.tensor(ct, device="cuda"),
                                          (True,) * 6, flag)
                qpmod.FORCE_FLAGS = 0
                torch.cuda.synchronize()
                def dev_of(x, ref, m, rtol, atol):
                    x = x.cpu().numpy()[m]; ref = ref[m]
                    return float(np.max(np.abs(x - ref) / (atol + rtol * np.abs(ref)))) if x.size else 0.0
                devs = {"zhat": dev_of(zhat, o["zhat"], cm, 1e-6, 1e-8), "lam": dev_of(lam, o["lam"], cm, 1e-5, 1e-7),
                        "nu": dev_of(nu, o["nu"], cm, 1e-5, 1e-7), "slack": dev_of(slack, o["slack"], cm, 1e-5, 1e-7)}
                for k, t in zip("QpGhAb", gr):
                    devs["d" + k] = dev_of(t, og["d" + k], gm, 1e-4, 1e-6)
                st = int((info[:, 0] != 0).sum()); itmean = float(info[:, 1].float().mean())
                for k, v in devs.items():
                    key = (fam, k)
                    if v > worst.get(key, (0,))[0]:
                        worst[key] = (v, (nz, nineq, neq), kind, seed)
                    if v > 1.0:
                        bad += 1
                        print("EXCEEDS", fam, k, "%.2f x tolerance" % v, (nz, nineq, neq), kind, seed, flush=True)
            print("size", (nz, nineq, neq), kind, "seed", seed, "converged %.4f" % cm.mean(), "status!=0:", st, "iters mean %.2f" % itmean,
                  "%.0fs" % (time.time() - t0), flush=True)
print("worst deviation / tolerance per output:")
for k in sorted(worst):
    print("  %-10s %-6s %.3f  at %s" % (k[0], k[1], worst[k][0], worst[k][1:]))
sys.exit(1 if bad else 0)
