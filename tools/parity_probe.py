#!/usr/bin/env python3
"""Finds the problems of a stress_parity batch whose GPU result misses the oracle's by more than
the test tolerances and dumps, for each of them, the oracle's residual history and the best
iterate both sides return when the solve is truncated at max_iter = 1..20 -- i.e. which exit
rule fired and at which mu -- to gpurun_out/probe/*.npz for offline reading."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle
from diff_qp_mpc_amd import qp as qpmod, _lib

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from families import family, family_mpc

oracle.build()")], {"np": np, "torch": torch}, ns)
family, family_mpc = ns["family"], ns["family_mpc"]

oracle.build()
out = os.path.join("gpurun_out", "probe")
os.makedirs(out, exist_ok=True)
B = int(os.environ.get("BATCH", "2048"))
cases = [("M", 0), ("M", 1), ("D", 1), ("D", 3)]
for kind, seed in cases:
    nz, nineq, neq = 30, 30, 15
    ins = family_mpc(seed, B) if kind == "M" else family(1000 * seed + nz, B, nz, nineq, neq, kind)
    o = oracle.qp_forward(*ins)
    dv = [torch.tensor(a, device="cuda") for a in ins]
    ct = np.random.default_rng(seed).standard_normal((B, nz))
    og = oracle.qp_backward(ins[0], ins[2], ins[4], o["zhat"], o["lam"], o["nu"], o["slack"], ct)
    zhat, lam, nu, slack, info, resid, saved = qpmod._forward_impl(*dv, 1e-12, 20, 3)
    gr = qpmod._backward_impl(saved, zhat, lam, nu, slack, torch.tensor(ct, device="cuda"), (True,) * 6, 0)
    torch.cuda.synchronize()
    cm = o["best_resid"] < 1e-8
    gm = cm & (np.maximum(o["lam"], o["slack"]).min(1) > 1e-5)
    def dev(x, ref, rtol, atol):
        e = np.abs(x.cpu().numpy() - ref) / (atol + rtol * np.abs(ref))
        return e.reshape(B, -1).max(1)
    e_f = np.maximum.reduce([dev(zhat, o["zhat"], 1e-6, 1e-8), dev(lam, o["lam"], 1e-5, 1e-7),
                             dev(slack, o["slack"], 1e-5, 1e-7)]) * cm
    e_g = np.maximum.reduce([dev(t, og["d" + k], 1e-4, 1e-6) for k, t in zip("QpGhAb", gr)]) * gm
    badi = np.nonzero((e_f > 1) | (e_g > 1))[0]
    print(kind, seed, "bad problems:", badi, "fwd dev", e_f[badi], "grad dev", e_g[badi], flush=True)
    if not len(badi):
        continue
    # truncated solves on both sides
    rec = dict(idx=badi, hist=o["resid_hist"][badi], o_lam=o["lam"][badi], o_slack=o["slack"][badi],
               o_zhat=o["zhat"][badi], o_iters=o["iters"], g_lam=lam.cpu().numpy()[badi],
               g_slack=slack.cpu().numpy()[badi], g_zhat=zhat.cpu().numpy()[badi],
               g_info=info.cpu().numpy()[badi], g_resid=resid.cpu().numpy()[badi],
               e_f=e_f[badi], e_g=e_g[badi])
    gk_res, gk_it, gk_lam, gk_sl, ok_lam, ok_sl, ok_res = [], [], [], [], [], [], []
    for k in range(1, 21):
        z2, l2, n2, s2, i2, r2, _ = qpmod._forward_impl(*dv, 1e-12, k, 3)
        torch.cuda.synchronize()
        gk_res.append(r2.cpu().numpy()[badi]); gk_it.append(i2.cpu().numpy()[badi, 1])
        gk_lam.append(l2.cpu().numpy()[badi]); gk_sl.append(s2.cpu().numpy()[badi])
        ok = oracle.qp_forward(*ins, maxIter=k)
        ok_lam.append(ok["lam"][badi]); ok_sl.append(ok["slack"][badi]); ok_res.append(ok["best_resid"][badi])
    rec.update(gk_res=np.array(gk_res), gk_it=np.array(gk_it), gk_lam=np.array(gk_lam), gk_sl=np.array(gk_sl),
               ok_lam=np.array(ok_lam), ok_sl=np.array(ok_sl), ok_res=np.array(ok_res))
    for j, i in enumerate(badi):
        rec["in%d_" % j + "Q"] = ins[0][i]; rec["in%d_p" % j] = ins[1][i]; rec["in%d_G" % j] = ins[2][i]
        rec["in%d_h" % j] = ins[3][i]; rec["in%d_A" % j] = ins[4][i]; rec["in%d_b" % j] = ins[5][i]
        rec["ct%d" % j] = ct[i]
    np.savez(os.path.join(out, "%s_%d.npz" % (kind, seed)), **rec)
    for j, i in enumerate(badi):
        print(" problem", i, "gpu iters", rec["g_info"][j], "gpu best", rec["g_resid"][j])
        print("  oracle hist", " ".join("%.2e" % v for v in rec["hist"][j]))
        print("  gpu best@k ", " ".join("%.2e" % v for v in rec["gk_res"][:, j]))
        print("  ora best@k ", " ".join("%.2e" % v for v in rec["ok_res"][:, j]))
