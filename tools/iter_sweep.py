#!/usr/bin/env python3
"""Forward kernel time as a function of max_iter (production library, HIP events): the intercept is
setup + epilogue, the slope the cost of one PDIPM iteration when every wavefront runs it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
dev = torch.device("cuda", 0)
B = int(os.environ.get("BATCH", "4096"))
ins = bench.family_R(0, B, 30, 30, 15)
for nullspace in (True, False):
    res = []
    for mi in (1, 2, 4, 8, 12, 16, 20):
        hp = bench.HotPath(dev, ins, termination="per_problem")
        hp.opts.max_iter = mi
        hp.opts.eps = 0.0; hp.opts.stall_tol = 0.0      # nobody exits early
        if not nullspace:
            hp.wsp = hp.null
        for _ in range(3): hp.forward()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in ev:
            a.record(); hp.forward(); b.record()
        torch.cuda.synchronize()
        t = float(np.median([a.elapsed_time(b) for a, b in ev]))
        res.append((mi, t, float(hp.info[:, 1].float().mean())))
    sl = (res[-1][1] - res[0][1]) / (res[-1][0] - res[0][0])
    print("nullspace" if nullspace else "rows", "B", B, " ".join("%d:%.4f" % (m, t) for m, t, _ in res),
          " per-iteration %.2f us, setup+epilogue %.1f us" % (sl * 1e3, (res[0][1] - sl) * 1e3), "iters", res[-1][2])
