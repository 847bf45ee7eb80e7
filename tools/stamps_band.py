#!/usr/bin/env python3
"""Per-phase s_memtime split of the block-tridiagonal Newton kernel's forward sweep (large models: the quadrotor) --
instrumented build of csrc/dqp_al_banded.hip (-DDQP_BAND_STAMPS), config-4 shape.
    python tools/ab_variants.py build dqp_al_banded.hip stamps=diff-qp-mpc_amd/csrc/dqp_al_banded.hip,-DDQP_BAND_STAMPS   (CPU box)
    DQP_HIP_LIBRARY=diff-qp-mpc_amd/csrc/libdqp_hip_ab_stamps.so python tools/stamps_band.py                               (GPU box)
Every stamp drains the memory counters: phases do not overlap as in the shipped kernel."""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from diff_qp_mpc_amd import _lib

lib = _lib.load()
lib.dqp_debug_band_stamps.argtypes = [ctypes.c_void_p]
args = argparse.Namespace(batch=None, graph=False, robot=None, T=None)
wl = bench.QuadrotorAL(torch, torch.device("cuda:0"), 0, 1, args)
for _ in range(2): wl.step()
torch.cuda.synchronize()
nwg = (wl.B + 3) // 4
st = torch.zeros(nwg, 8, dtype=torch.int64, device="cuda")
lib.dqp_debug_band_stamps(ctypes.c_void_p(st.data_ptr()))
wl.step(); torch.cuda.synchronize()          # the last Newton launch of the step leaves its stamps
lib.dqp_debug_band_stamps(ctypes.c_void_p(0))
s = st.cpu().numpy().astype(np.float64)
names = ["knot loads (+ previous stores)", "model + Jacobian column", "gradient, H (J^T J, M^T M)", "Cholesky",
         "forward substitutions, stores", "backward sweep"]
tot = s[:, :6].sum(1)
print("per wavefront, 30 knots, s_memtime ticks (core clock here): total %.0f (min %.0f max %.0f)" % (tot.mean(), tot.min(), tot.max()))
for k, nm in enumerate(names):
    print("  %-34s %8.1f  (%4.1f %%)   min %7.0f  max %7.0f" % (nm, s[:, k].mean(), 100 * s[:, k].mean() / tot.mean(), s[:, k].min(), s[:, k].max()))
