#!/usr/bin/env python3
"""Wide parity sweep on the GPU box: many seeds x the DPP-row sizes x both forward families,
GPU (through the Python mirror -> C ABI) against the CPU oracle, per problem.  Problems the
oracle itself did not converge on (best residual >= 1e-8) are masked, as in the tests.
Samples on which the literal reference (unguarded get_step, batch.py:211-214) and the guarded one (its own
batch_LU.get_step) DIFFER beyond the tolerances are compared with the guarded oracle, all others with the literal
one (DESIGN.md "Parity"; tools/strict_vs_guard.py counts them).  Prints the worst deviations; exits non-zero if any exceeds the test tolerances."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle
from diff_qp_mpc_amd import qp as qpmod, _lib, _build

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from families import family, family_mpc

oracle.build()
qpmod.STALL_TOL = float(os.environ.get("STALL_TOL", qpmod.STALL_TOL))
seeds = int(os.environ.get("SEEDS", "6"))
B = int(os.environ.get("BATCH", "2048"))
worst = {}
bad = 0
nbd = ntot = 0
t0 = time.time()
for (nz, nineq, neq) in _build.R16N_SIZES:
    for kind in os.environ.get("KINDS", "R,D,M").split(","):
        if kind == "M" and (nz, nineq, neq) != (30, 30, 15):
            continue
        for seed in range(seeds):
            ins = family_mpc(seed, B) if kind == "M" else family(1000 * seed + nz, B, nz, nineq, neq, kind)
            o = oracle.qp_forward(*ins)                      # literal batch.py
            o2 = oracle.qp_forward(*ins, guard=True)         # with batch_LU.get_step
            bd = np.zeros(B, dtype=bool)                    # samples where the two actually differ beyond tolerance
            for k, (rt, at) in (("zhat", (1e-6, 1e-8)), ("lam", (1e-5, 1e-7)), ("nu", (1e-5, 1e-7)), ("slack", (1e-5, 1e-7))):
                e = ~(np.abs(o[k] - o2[k]) <= at + rt * np.abs(o2[k]))
                bd |= e.reshape(B, -1).any(1)
            for k in ("zhat", "lam", "nu", "slack", "best_resid"):
                o[k][bd] = o2[k][bd]
            nbd += int(bd.sum()); ntot += B
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
print("samples on which the literal and the guarded reference differ (compared with the guarded one): %d of %d" % (nbd, ntot))
print("worst deviation / tolerance per output:")
for k in sorted(worst):
    print("  %-10s %-6s %.3f  at %s" % (k[0], k[1], worst[k][0], worst[k][1:]))
sys.exit(1 if bad else 0)
