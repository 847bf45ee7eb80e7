"""Where the non-solver time of a config-5 training step goes: torch.profiler table of one step (sorted by device time)."""
import argparse, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench

ap = argparse.ArgumentParser(); ap.add_argument("--blas", default=None); ap.add_argument("--batch", type=int, default=None)
a = ap.parse_args()
if a.blas:
    torch.backends.cuda.preferred_blas_library(a.blas)
dev = torch.device("cuda:0")
args = argparse.Namespace(batch=a.batch, graph=False, robot=None, T=None)
wl = bench.DEQMPCTrain(torch, dev, 0, 1, args)
for _ in range(2):
    wl.step()
torch.cuda.synchronize()
import time
t = time.perf_counter()
for _ in range(5):
    wl.step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t) / 5 * 1e3)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    wl.step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
