import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, ctypes
from diff_qp_mpc_amd import qp as qpmod, _lib
from families import family
B = 128
for nz in (100, 500):
    ins = [torch.tensor(a, device="cuda") for a in family(1, B, nz, nz, 0, "R")]
    z, l, n, s, info, r, saved = qpmod._forward_impl(*ins, 1e-12, 20, 3, termination="per_problem")
    torch.cuda.synchronize()
    ws = saved[4]
    NP = (nz + 63) // 64 * 64
    per = ws.numel() // B
    # oProf = total - 16
    prof = ws.view(B, per)[:, per - 16: per - 8].cpu().numpy()
    sub = ws.view(B, per)[:, per - 8: per].cpu().numpy()
    its = info[:, 1].float().mean().item()
    names = ["setup", "residuals", "factor_T", "kkt_wz(affine)", "corrector trsv", "kkt_xy+update"]
    tot = prof[:, :6].sum(1).mean()
    print("nz", nz, "iters %.1f" % its, " total Mcycles %.2f (%.2f ms @2.1GHz)" % (tot / 1e6, tot / 2.1e6))
    for k, nm in enumerate(names):
        print("   %-18s %8.3f Mcycles  %5.1f%%   per iteration %.1f kcycles" % (nm, prof[:, k].mean() / 1e6, 100 * prof[:, k].mean() / tot, prof[:, k].mean() / 1e3 / (1 if k == 0 else its)))
    for k, nm in enumerate(["potrf: tile_mm (update)", "potrf: C -> LDS", "potrf: chol64", "potrf: inv64", "potrf: diag writes", "potrf: C dinv^T + write"]):
        print("      %-26s per iteration %.1f kcycles" % (nm, sub[:, k].mean() / 1e3 / its))
