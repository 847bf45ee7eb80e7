import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
dev = torch.device("cuda", 0)
hp = bench.HotPath(dev, bench.family_R(0, 4096, 30, 30, 15))
hp.forward(); torch.cuda.synchronize()
it = hp.info[:, 1].cpu().numpy()
print("hist", np.bincount(it, minlength=21))
w = it.reshape(-1, 4).max(1)
print("per-wave max hist", np.bincount(w, minlength=21), "mean of wave max", w.mean())
br = hp.resid.cpu().numpy()
if br is not None:
    for k in (17, 18, 19, 20):
        sel = it == k
        if sel.any(): print(k, "best resid quantiles", np.quantile(br[sel], [0, .5, 1]))
