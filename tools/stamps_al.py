#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime split of al_newton_kernel at the cartpole T=20 shape."""
import ctypes, os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from diff_qp_mpc_amd import _lib, _build

# instrumented variant: only dqp_al.hip is recompiled with -DDQP_STAMPS (run once on the CPU box
# first: `python tools/stamps_al.py --build`; the .so travels with gpurun)
so = os.path.join(_build.CSRC, "libdqp_hip_stamps_al.so")
if "--build" in sys.argv or not os.path.exists(so):
    objs = []
    for obj, cmd in _build._jobs():
        if "dqp_al" in obj:
            o2 = obj.replace(".o", ".stamps.o")
            subprocess.check_call([c if c != obj else o2 for c in cmd] + ["-DDQP_STAMPS"])
            obj = o2
        objs.append(obj)
    subprocess.check_call([_build.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
    if "--build" in sys.argv:
        sys.exit(0)
_build.SO = so
lib = _lib.load()
lib.dqp_debug_set_stamps.argtypes = [ctypes.c_void_p]
from diff_qp_mpc_amd import al_utils
B, nz, ncon = 4096, int(os.environ.get("NZ", 100)), int(os.environ.get("NCON", 120))
g = torch.Generator(device="cuda").manual_seed(0)
J = torch.randn(B, ncon, nz, dtype=torch.float64, device="cuda", generator=g)
Qd = torch.rand(B, nz, dtype=torch.float64, device="cuda", generator=g) + 0.01
rho = torch.full((B, 1), 10.0, dtype=torch.float64, device="cuda")
grad = torch.randn(B, nz, dtype=torch.float64, device="cuda", generator=g)
terms = al_utils.HessianTerms(J, Qd, rho)
al_utils.newton_step(terms, grad); torch.cuda.synchronize()
st = torch.zeros(B, 16, dtype=torch.int64, device="cuda")
lib.dqp_debug_set_stamps(ctypes.c_void_p(st.data_ptr()))
al_utils.newton_step(terms, grad); torch.cuda.synchronize()
lib.dqp_debug_set_stamps(ctypes.c_void_p(0))
s = st.cpu().numpy().astype(np.float64)
names = ["stage Jc + MFMA", "write H", "cholesky", "scale", "wave solve (waves 1-3 store L)"]
for k, nm in enumerate(names):
    d = s[:, k + 1] - s[:, k]
    print("%-18s median %9.0f  min %9.0f  max %9.0f (s_memtime ticks ~ shader cycles)" % (nm, np.median(d), d.min(), d.max()))
print("total median", np.median(s[:, 5] - s[:, 0]))
