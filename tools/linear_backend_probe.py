"""Which BLAS backend torch picks for the DEQLayer's forward linears at config-5 size (B = 65536, fp32), and what
each costs: the hipBLASLt heuristic's choice for x @ W^T + b at (65536 x 128) x (128 x 128) was 226 us per call in
profiles/r3/config5_kernel_stats.csv (a 32x32x256 macro tile), ten times the backward GEMMs of the same size."""
import sys, time
import torch
import torch.nn.functional as F

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
shapes = [(35, 128), (128, 128), (128, 28)]
for lib in ("default", "cublaslt", "cublas"):
    if lib != "default":
        torch.backends.cuda.preferred_blas_library(lib)
    for (k, n) in shapes:
        x = torch.randn(B, k, device="cuda"); w = torch.randn(n, k, device="cuda"); b = torch.randn(n, device="cuda")
        variants = {"linear": lambda: F.linear(x, w, b), "addmm": lambda: torch.addmm(b, x, w.t()), "mm+b": lambda: torch.mm(x, w.t()) + b}
        for name, fn in variants.items():
            for _ in range(5): fn()
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(50): fn()
            torch.cuda.synchronize()
            print("%-9s k=%3d n=%3d %-7s %8.1f us" % (lib, k, n, name, (time.perf_counter() - t) / 50 * 1e6), flush=True)
