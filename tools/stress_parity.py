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

def family_mpc(seed, B, n=3, m=3, T=5):
    """MPC-structured dense QP (SURVEY 8d family M): block-diagonal cost, dynamics equalities
    x_{t+1} = A x_t + B u_t, x_0 given, box |u| <= 1 (so many constraints are active)."""
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


oracle.build()
qpmod.STALL_TOL = float(os.environ.get("STALL_TOL", qpmod.STALL_TOL))
seeds = int(os.environ.get("SEEDS", "6"))
B = int(os.environ.get("BATCH", "2048"))
worst = {}
bad = 0
t0 = time.time()
for (nz, nineq, neq) in _build.R16N_SIZES:
    for kind in os.environ.get("KINDS", "R,D,M").split(","):
        if kind == "M" and (nz, nineq, neq) != (30, 30, 15):
            continue
        for seed in range(seeds):
            ins = family_mpc(seed, B) if kind == "M" else family(1000 * seed + nz, B, nz, nineq, neq, kind)
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
                nexc = [0]
                def dev_of(x, ref, m, rtol, atol):
                    x = x.cpu().numpy()[m]; ref = ref[m]
                    if not x.size:
                        return 0.0
                    e = np.abs(x - ref) / (atol + rtol * np.abs(ref))
                    nexc[0] = max(nexc[0], int((e.reshape(e.shape[0], -1).max(1) > 1.0).sum()))
                    return float(e.max())
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
                if nexc[0]:
                    print("   problems over tolerance in this batch (%s): %d of %d" % (fam, nexc[0], B), flush=True)
            print("size", (nz, nineq, neq), kind, "seed", seed, "converged %.4f" % cm.mean(), "status!=0:", st, "iters mean %.2f" % itmean,
                  "%.0fs" % (time.time() - t0), flush=True)
print("worst deviation / tolerance per output:")
for k in sorted(worst):
    print("  %-10s %-6s %.3f  at %s" % (k[0], k[1], worst[k][0], worst[k][1:]))
sys.exit(1 if bad else 0)
