#!/usr/bin/env python3
"""Diagnostic: per-phase cycle split of the DPP-row forward kernel (s_memtime stamps)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import subprocess
import bench
from diff_qp_mpc_amd import _lib, _build

# build the instrumented variant next to the product library (never loaded by the product)
so = os.path.join(_build.CSRC, "libdqp_hip_stamps.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(_build.SO):
    objs, procs = [], []
    for obj, cmd in _build._jobs():
        # only the metric-size forward kernels and the host file carry stamps; every other object
        # is the product's own
        if os.path.basename(obj) in ("dqp_r16n_30_30_15.o", "dqp_r16f_30_30_15.o", "dqp_pdipm.o", "dqp_al.o"):
            o2 = obj.replace(".o", ".stamps.o")
            procs.append(subprocess.Popen([c if c != obj else o2 for c in cmd] + ["-DDQP_STAMPS"]))
            objs.append(o2)
        else:
            objs.append(obj)
    for pr in procs:
        assert pr.wait() == 0
    subprocess.check_call([_build.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
_build.SO = so
lib = _lib.load()
lib.dqp_debug_set_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda", 0)
hp = bench.HotPath(dev, bench.family_R(0, int(os.environ.get("BATCH", "4096")), 30, 30, 15),
                   termination=os.environ.get("TERMINATION", "batch"))
if os.environ.get("NULLSPACE", "1") != "1":
    hp.wsp = hp.null
hp.forward(); hp.backward(); torch.cuda.synchronize()
nb = 1024
st = torch.zeros(nb, 16, dtype=torch.int64, device=dev)
lib.dqp_debug_set_stamps(ctypes.c_void_p(st.data_ptr()))
hp.forward(); torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.float64)
lib.dqp_debug_set_stamps(ctypes.c_void_p(0))
if os.environ.get("NULLSPACE", "1") == "1":
    names = {1: "A load Q+chol+store", 2: "B G,A rows * Lq^-T", 3: "C Householder LQ + xy", 4: "E,F ph, hp, w1",
             5: "G LqZ, qv", 15: "H park ctx + Gz Gz^T", 6: "init factor+solve+gz"}
    order = [1, 2, 3, 4, 5, 15, 6]
else:
    names = {1: "A load Q+chol+store", 2: "B G,A rows * Lq^-T", 3: "C S11,chol,At", 4: "W", 5: "D R",
             6: "p^,b~, init factor+solve"}
    order = list(range(1, 7))
prev = s[:, 0]
for k in order:
    d = s[:, k] - prev
    print("%-28s median %8.0f  min %8.0f max %8.0f  (s_memtime ticks)" % (names[k], np.median(d), d.min(), d.max()))
    prev = s[:, k]
print("iteration 0 total            median %8.0f" % np.median(s[:, 8] - s[:, 6]))
print("iter1 residuals+termination   median %8.0f" % np.median(s[:, 9] - s[:, 8]))
it = [("residuals+resid", 9, None), ("factor_T", 10, 9), ("affine solve", 11, 10), ("corrector solve", 12, 11), ("xy + step", 13, 12)]
for name, k, p in it[1:]:
    print("iter1 %-22s median %8.0f" % (name, np.median(s[:, k] - s[:, p])))
print("loop total (all iterations)  median %8.0f  max %8.0f" % (np.median(s[:, 7] - s[:, 6]), (s[:, 7] - s[:, 6]).max()))
print("epilogue                     median %8.0f" % np.median(s[:, 14] - s[:, 7]))
print("kernel total                 median %8.0f  max %8.0f" % (np.median(s[:, 14] - s[:, 0]), (s[:, 14] - s[:, 0]).max()))
print("iters: mean %.2f max %d" % (hp.info[:, 1].float().mean().item(), hp.info[:, 1].max().item()))
