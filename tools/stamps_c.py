#!/usr/bin/env python3
"""Experiment: cycles per Householder reflector in the null-space setup (DQP_STAMPS_C build)."""
import ctypes, os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from diff_qp_mpc_amd import _lib, _build
so = os.path.join(_build.CSRC, "libdqp_hip_stamps_c.so")
if "--build" in sys.argv:
    objs, procs = [], []
    for obj, cmd in _build._jobs():
        if os.path.basename(obj) in ("dqp_r16n_30_30_15.o", "dqp_pdipm.o", "dqp_al.o"):
            o2 = obj.replace(".o", ".stampsc.o")
            extra = ["-DDQP_STAMPS"] + (["-DDQP_STAMPS_C"] if "r16n" in obj else [])
            procs.append(subprocess.Popen([c if c != obj else o2 for c in cmd] + extra))
            objs.append(o2)
        else:
            objs.append(obj)
    for pr in procs:
        assert pr.wait() == 0
    subprocess.check_call([_build.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
    sys.exit(0)
_build.SO = so
lib = _lib.load()
lib.dqp_debug_set_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda", 0)
B = int(os.environ.get("BATCH", "4096"))
hp = bench.HotPath(dev, bench.family_R(0, B, 30, 30, 15), termination="batch")
hp.forward(); torch.cuda.synchronize()
st = torch.zeros((B + 3) // 4, 16, dtype=torch.int64, device=dev)
lib.dqp_debug_set_stamps(ctypes.c_void_p(st.data_ptr()))
hp.forward(); torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.float64)
for k in range(14, -1, -1):
    nxt = s[:, k - 1] if k > 0 else s[:, 15]
    d = nxt - s[:, k]
    print("reflector %2d: median %8.0f min %8.0f max %8.0f" % (k, np.median(d), d.min(), d.max()))
