import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from diff_qp_mpc_amd import qp as qpmod, _lib
import bench
B = 64
for name in ("dup_A_rows", "zero_G_row", "scaled_1e6", "scaled_1e-6", "infeasible", "nan_input", "Q_indef"):
    Q, p, G, h, A, b = [t.clone() for t in bench.family_R(5, B, 30, 30, 15)]
    if name == "dup_A_rows": A[:, 1] = A[:, 0]; b[:, 1] = b[:, 0]
    if name == "zero_G_row": G[:, 3] = 0; h[:, 3] = 1.0
    if name == "scaled_1e6": Q *= 1e6; p *= 1e6
    if name == "scaled_1e-6": Q *= 1e-6; p *= 1e-6
    if name == "infeasible": G[:, 1] = -G[:, 0]; h[:, 0] = -1.0; h[:, 1] = -1.0
    if name == "nan_input": p[3, 2] = float("nan")
    if name == "Q_indef": Q[7] = -Q[7]
    for fam, flag in (("nullspace", 0), ("rows", _lib.DQP_FLAG_NO_NULLSPACE), ("generic", _lib.DQP_FLAG_GENERIC_ONLY)):
        qpmod.FORCE_FLAGS = flag
        ins = [t.cuda() for t in (Q, p, G, h, A, b)]
        zhat, lam, nu, slack, info, resid, saved = qpmod._forward_impl(*ins, 1e-12, 20, 3)
        gr = qpmod._backward_impl(saved, zhat, lam, nu, slack, torch.ones_like(zhat), (True,) * 6, flag)
        torch.cuda.synchronize()
        qpmod.FORCE_FLAGS = 0
        fin = torch.isfinite(zhat).all(1)
        print("%-12s %-9s finite zhat %3d/%d  status counts %s  iters max %d  best_resid median %.1e  grads finite %s"
              % (name, fam, int(fin.sum()), B, np.bincount(info[:, 0].cpu().numpy(), minlength=3).tolist(),
                 int(info[:, 1].max()), float(resid.nanmedian()), bool(all(torch.isfinite(g).all() for g in gr if g is not None))), flush=True)
