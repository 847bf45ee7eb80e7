#!/usr/bin/env python3
"""Forward / backward kernel time vs batch size at the metric shape (is the chip filled?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
for B in (1024, 2048, 4096, 8192, 16384, 32768):
    hp = bench.HotPath(dev, bench.family_R(0, B, 30, 30, 15))
    for _ in range(3):
        hp.forward(); hp.backward()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    reps = 10
    for _ in range(reps):
        ev[0].record(); hp.forward(); ev[1].record(); hp.backward(); ev[2].record()
        torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
    print("B=%6d  forward %.3f ms  backward %.3f ms  -> %.2f M QP/s" % (B, tf / reps, tb / reps, B / (tf + tb) * reps / 1e3), flush=True)
