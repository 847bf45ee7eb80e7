#!/usr/bin/env python3
"""Forward/backward kernel times of the BASELINE.json config shapes through the Python mirror's
implementation functions (no autograd overhead): which kernel family serves which shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from diff_qp_mpc_amd import qp as qpmod

shapes = [("config 0/metric  n3 m3 T5", 4096, 30, 30, 15), ("pendulum MPC    n3 m1 T5", 4096, 20, 10, 15),
          ("config 2 shape  n3 m1 T10", 1024, 40, 20, 30), ("cartpole-2      n6 m1 T5", 4096, 35, 10, 30),
          ("cartpole-1      n4 m1 T5", 4096, 25, 10, 20), ("64-dim limit", 1024, 64, 64, 32)]
for name, B, nz, nineq, neq in shapes:
    ins = [t.cuda() for t in bench.family_R(0, B, nz, nineq, neq)]
    out = qpmod._forward_impl(*ins, 1e-12, 20, 3)
    ct = torch.ones(B, nz, dtype=torch.float64, device="cuda")
    def fwd(): return qpmod._forward_impl(*ins, 1e-12, 20, 3)
    def bwd(o): return qpmod._backward_impl(o[6], o[0], o[1], o[2], o[3], ct, (True,) * 6, 0)
    for _ in range(2): bwd(fwd())
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0; reps = 5
    for _ in range(reps):
        e[0].record(); o = fwd(); e[1].record(); bwd(o); e[2].record(); torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    print("%-28s B=%5d nz=%2d nineq=%2d neq=%2d  forward %.3f ms  backward %.3f ms  %.2f M QP/s  iters %.1f"
          % (name, B, nz, nineq, neq, tf / reps, tb / reps, B / (tf + tb) * reps / 1e3, float(o[4][:, 1].float().mean())), flush=True)
