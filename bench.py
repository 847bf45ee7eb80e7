#!/usr/bin/env python3
"""Benchmark of the hot path: QPs/sec for one forward + one backward of the differentiable
batched QP solver on the BASELINE.json metric config (batch=4096, n_state=3, n_ctrl=3, T=5 ->
nz=30, nineq=30, neq=15; random dense family R of SURVEY.md §8d), fp64.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path over one batch resident in HBM: dqp_qp_forward followed
by dqp_qp_backward (cotangent = ones), called through the C ABI on torch's current stream.
With N > 1 (launched by torch.distributed.run, one rank per GPU) every rank solves its own
4096-QP shard (weak scaling, no data-path collective in the timed region except the single
all_gather of the solved zhat batch that BASELINE.json's north_star names).

The headline runs in the parity-safe mode (DQP_FLAG_BATCH_TERMINATION: the reference's
batch-coupled stopping rule replayed on the device, the default of the Python mirrors); the
per-problem-exit mode is timed afterwards and reported as `fast_mode`.

Prints ONE JSON line (rank 0) with `roofline` (the dominant kernel = pass 1 of the forward, timed
alone with HIP events on the launch stream; algorithmic bytes of SURVEY.md §8d), `cpu_baseline`
(the C oracle, a port of the reference algorithm, timed on this host's cores on the same workload)
and `fast_mode`.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

B_PER_GPU = 4096
NZ, NINEQ, NEQ = 30, 30, 15
# algorithmic elements per QP
FWD_ELEMS = (NZ * NZ + NZ + NINEQ * NZ + NINEQ + NEQ * NZ + NEQ) + (NZ + 2 * NINEQ + NEQ)   # 2325 + 105
BWD_ELEMS = (NZ * NZ + NINEQ * NZ + NEQ * NZ + NZ + 2 * NINEQ + NEQ + NZ) + \
            (NZ * NZ + NZ + NINEQ * NZ + NINEQ + NEQ * NZ + NEQ)                             # 2385 + 2325
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6      # public MI355X spec (vector = matrix fp64); not in the guide
# ~50 kFLOP per PDIPM iteration + 0.27 MFLOP one-time factorisations per QP
FLOP_SETUP, FLOP_PER_ITER = 0.27e6, 50e3
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r2", "final_pmc_summary.json")


def library_fingerprint():
    """sha1 over the sources of the kernels this command runs (dense QP forward / backward, batch-rule
    reduction): a PMC summary is only quoted for the kernels it was taken on."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "diff-qp-mpc_amd", "csrc")
    for f in ("dqp_common.h", "dqp_r16_prims.h", "dqp_r16n.hip", "dqp_r16.hip", "dqp_term.hip", "dqp_pdipm.hip",
              "dqp_dispatch.hip"):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc(kernel_key):
    """Counters of the committed rocprofv3 --pmc passes of this same command (tools/profile_round.sh;
    FETCH_SIZE and WRITE_SIZE in separate passes, in KB; gfx950 FETCH_SIZE counts half of the
    fetched bytes for wide streaming reads, MI355X_MICROARCH.md §HBM).  Empty when there is no
    summary or it was taken on different kernel sources (then `traffic` is reported as null)."""
    try:
        d = json.load(open(PMC_SUMMARY))
        if d.get("_library_fingerprint") != library_fingerprint():
            return {}
        return next(v for n, v in d.items() if kernel_key in n)
    except Exception:
        return {}


def measured_traffic(kernel_key):
    k = pmc(kernel_key)
    if "FETCH_SIZE" not in k or "WRITE_SIZE" not in k:
        return None
    return (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0


def family_R(seed, B, nz, nineq, neq):
    g = torch.Generator().manual_seed(seed)
    L = torch.randn(B, nz, nz, generator=g, dtype=torch.float64)
    Q = L @ L.transpose(1, 2) + 1e-3 * torch.eye(nz, dtype=torch.float64)
    G = torch.randn(B, nineq, nz, generator=g, dtype=torch.float64)
    z0 = torch.randn(B, nz, generator=g, dtype=torch.float64)
    s0 = torch.rand(B, nineq, generator=g, dtype=torch.float64)
    A = torch.randn(B, neq, nz, generator=g, dtype=torch.float64)
    p = torch.randn(B, nz, generator=g, dtype=torch.float64)
    h = (G @ z0.unsqueeze(-1)).squeeze(-1) + s0
    b = (A @ z0.unsqueeze(-1)).squeeze(-1)
    return Q, p, G, h, A, b


class HotPath:
    """Pre-allocated buffers + the two C-ABI calls of one step."""

    def __init__(self, dev, host_inputs, termination="batch"):
        from diff_qp_mpc_amd import _lib
        self._lib = _lib
        self.lib = _lib.load()
        self.dev = dev
        B = host_inputs[0].shape[0]
        self.B = B
        self.Q, self.p, self.G, self.h, self.A, self.b = [t.to(dev).contiguous() for t in host_inputs]
        kw = dict(dtype=torch.float64, device=dev)
        self.zhat = torch.empty(B, NZ, **kw); self.lam = torch.empty(B, NINEQ, **kw)
        self.nu = torch.empty(B, NEQ, **kw); self.slack = torch.empty(B, NINEQ, **kw)
        self.info = torch.empty(B, 2, dtype=torch.int32, device=dev)
        self.resid = torch.empty(B, **kw)
        self.ct = torch.ones(B, NZ, **kw)
        self.dQ = torch.empty(B, NZ, NZ, **kw); self.dp = torch.empty(B, NZ, **kw)
        self.dG = torch.empty(B, NINEQ, NZ, **kw); self.dh = torch.empty(B, NINEQ, **kw)
        self.dA = torch.empty(B, NEQ, NZ, **kw); self.db = torch.empty(B, NEQ, **kw)
        self.dims = _lib.dqp_dims(B, NZ, NINEQ, NEQ, NZ * NZ, NZ, NINEQ * NZ, NINEQ, NEQ * NZ, NEQ)
        # "batch": the reference's batch-coupled stopping rule replayed on the device (parity-safe,
        # the package default); "per_problem": every QP stops on its own (include/dqp.h)
        tflag = _lib.DQP_FLAG_BATCH_TERMINATION if termination == "batch" else 0
        self.opts = _lib.dqp_opts(float(os.environ.get("DQP_BENCH_EPS", "1e-12")), 1e-10, 20, 3, tflag, 0)
        self.opts_pass1 = _lib.dqp_opts(self.opts.eps, 1e-10, 20, 3, tflag | _lib.DQP_FLAG_HISTORY_ONLY, 0)
        wsb = int(self.lib.dqp_workspace_bytes(ctypes.byref(self.dims)))
        self.ws = torch.empty(max(wsb // 8, 1), **kw)            # caller-owned scratch (include/dqp.h)
        tb = int(self.lib.dqp_termination_bytes(ctypes.byref(self.dims), ctypes.byref(self.opts)))
        self.term = torch.empty(max((tb + 7) // 8, 1), **kw)
        self.stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        self.fargs = [P(t) for t in (self.Q, self.p, self.G, self.h, self.A, self.b, self.zhat,
                                     self.lam, self.nu, self.slack, self.info, self.resid)]
        self.bargs = [P(t) for t in (self.Q, self.G, self.A, self.zhat, self.lam, self.nu,
                                     self.slack, self.ct, self.dQ, self.dp, self.dG, self.dh,
                                     self.dA, self.db)]
        self.null = ctypes.c_void_p(0)
        self.wsp = P(self.ws) if wsb > 0 else self.null
        self.termp = P(self.term) if tb > 0 else self.null
        # backward restarts from the factorisation context forward leaves in the workspace (what the
        # reference keeps on ctx, qp.py:93-95)
        self.bopts = self._lib.dqp_opts(0.0, 0.0, 0, 0, self._lib.DQP_FLAG_BACKWARD_CTX if wsb > 0 else 0, 0)

    def forward(self):
        rc = self.lib.dqp_qp_forward(ctypes.byref(self.dims), ctypes.byref(self.opts), *self.fargs,
                                     self.wsp, self.termp, self.stream)
        if rc:
            raise RuntimeError("dqp_qp_forward rc=%d" % rc)

    def forward_pass1(self):
        """the dominant launch alone (DQP_FLAG_HISTORY_ONLY), for the roofline figure"""
        rc = self.lib.dqp_qp_forward(ctypes.byref(self.dims), ctypes.byref(self.opts_pass1), *self.fargs,
                                     self.wsp, self.termp, self.stream)
        if rc:
            raise RuntimeError("dqp_qp_forward rc=%d" % rc)

    def backward(self):
        rc = self.lib.dqp_qp_backward(ctypes.byref(self.dims), ctypes.byref(self.bopts), *self.bargs,
                                      self.null, self.wsp, self.stream)
        if rc:
            raise RuntimeError("dqp_qp_backward rc=%d" % rc)


def cpu_baseline(host_inputs, reps=3):
    """The oracle (a C port of the reference's algorithm, OpenMP over the batch) on this host."""
    from oracle import oracle
    Q, p, G, h, A, b = [t.numpy() for t in host_inputs]
    B = Q.shape[0]
    nthreads = oracle.max_threads()
    ct = np.ones((B, NZ))
    ts = []
    for r in range(reps + 1):
        t0 = time.perf_counter()
        o = oracle.qp_forward(Q, p, G, h, A, b, nthreads=nthreads)
        oracle.qp_backward(Q, G, A, o["zhat"], o["lam"], o["nu"], o["slack"], ct, nthreads=nthreads)
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts[1:]))
    return {"value": B / t, "unit": "QPs/sec", "cores": nthreads, "kind": "port",
            "sample": "same workload, %d QPs fwd+bwd, median of %d reps after 1 warm-up; "
                      "batch-coupled PDIPM ran %d iterations" % (B, reps, o["iters"])}, o


def timed_steps(hp, steps, world, gathered):
    """K steps of forward + backward bracketed by barrier + synchronize; -> (elapsed s, fwd ms, bwd ms)."""
    fev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    bev = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        fev[k][0].record(); hp.forward(); fev[k][1].record()        # fev[k][1] also opens backward
        work = None
        if world > 1:   # north_star: a single RCCL gather of the solved batch, overlapped with backward
            work = dist.all_gather_into_tensor(gathered, hp.zhat, async_op=True)
        hp.backward(); bev[k].record()
        if work is not None:
            work.wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=hp.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    fwd_ms = float(np.mean([a.elapsed_time(b) for a, b in fev]))
    bwd_ms = float(np.mean([f[1].elapsed_time(b) for f, b in zip(fev, bev)]))
    return elapsed, fwd_ms, bwd_ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)       # 200 x 0.31 ms: a 60 ms timed region
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--termination", choices=["batch", "per_problem"], default="batch",
                    help="mode of the headline number (default: the parity-safe batch rule)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    # DQP_BENCH_BACKEND=gloo + DQP_BENCH_ONE_DEVICE=1 rehearse the N>1 code path with several
    # ranks on a single GPU (the 8-GPU run itself is the driver's job).
    one_dev = os.environ.get("DQP_BENCH_ONE_DEVICE") == "1"
    dev = torch.device("cuda", 0 if one_dev else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DQP_BENCH_BACKEND", "nccl")      # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    host_inputs = family_R(rank, B_PER_GPU, NZ, NINEQ, NEQ)
    hp = HotPath(dev, host_inputs, termination=args.termination)
    gathered = torch.empty(world * B_PER_GPU, NZ, dtype=torch.float64, device=dev) if world > 1 else None

    for _ in range(args.warmup):
        hp.forward()
        work = dist.all_gather_into_tensor(gathered, hp.zhat, async_op=True) if world > 1 else None
        hp.backward()
        if work is not None:
            work.wait()
    torch.cuda.synchronize()
    elapsed, fwd_ms, bwd_ms = timed_steps(hp, args.steps, world, gathered)
    iters = hp.info[:, 1].float()
    status_bad = int((hp.info[:, 0] != 0).sum())
    zhat_head = hp.zhat.clone()

    # the dominant launch alone (pass 1 of the batch rule = the forward kernel with every wavefront
    # at max_iter), HIP events on the launch stream
    pass1_ms = None
    if args.termination == "batch":
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for a, b in ev:
            a.record(); hp.forward_pass1(); b.record()
        torch.cuda.synchronize()
        pass1_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    # the other termination mode, same buffers and inputs
    other = "per_problem" if args.termination == "batch" else "batch"
    hp2 = HotPath(dev, host_inputs, termination=other)
    for _ in range(args.warmup):
        hp2.forward(); hp2.backward()
    torch.cuda.synchronize()
    elapsed2, fwd2_ms, bwd2_ms = timed_steps(hp2, args.steps, 1, None) if world == 1 else (None, None, None)

    if rank == 0:
        qps = world * B_PER_GPU * args.steps / elapsed
        fwd_bytes = FWD_ELEMS * 8 * B_PER_GPU
        bwd_bytes = BWD_ELEMS * 8 * B_PER_GPU
        kern_ms = pass1_ms if pass1_ms is not None else fwd_ms
        fwd_gbs = fwd_bytes / (kern_ms * 1e-3) / 1e9
        kpm = pmc("r16n::forward_kernel")
        traffic = measured_traffic("r16n::forward_kernel")
        out = {
            "metric": "QPs/sec (fwd+bwd), batch=4096 n=3 m=3 T=5",
            "value": qps, "unit": "QPs/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "random dense QP family R (SURVEY §8d), configs[0] shape at "
                                   "BASELINE metric batch: B=4096/GPU nz=30 nineq=30 neq=15",
                       "global_batch": world * B_PER_GPU, "n_state": 3, "n_ctrl": 3, "T": 5,
                       "parallelism": "batch-shard x%d" % world,
                       "termination": args.termination + (" (the reference's batch-coupled rule, parity-safe)"
                                                          if args.termination == "batch" else " (per-problem exit)")},
            "roofline": {"bound": "hbm", "kernel": "dqp::r16n::forward_kernel<Cfg<30,30,15>> (pass 1)",
                         "achieved": fwd_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": fwd_gbs / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": ("profiles/r2/final_pmc_summary.json (rocprofv3 --pmc passes of this "
                                            "command on these kernel sources)" if traffic is not None else None),
                         "avg_launch_ms": kern_ms, "algorithmic_bytes_per_launch": fwd_bytes,
                         "note": "fp64-issue bound, not HBM bound (DESIGN.md §4): every wavefront runs max_iter "
                                 "iterations at ~4.7 cycles per VALU instruction; measured traffic = algorithmic "
                                 "inputs/outputs + the factorisation context written for backward (which then "
                                 "reads it instead of Q, G, A) + the improving iterates and residual history the "
                                 "batch rule's finish pass reads"},
            "kernels": {"forward_call_ms": fwd_ms, "forward_pass1_kernel_ms": pass1_ms,
                        "qp_backward_kernel_ms": bwd_ms,
                        "backward_GBps": bwd_bytes / (bwd_ms * 1e-3) / 1e9,
                        "backward_hbm_frac": bwd_bytes / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "backward_traffic": measured_traffic("r16n::backward_kernel"),
                        "pdipm_iters_mean": float(iters.mean()), "pdipm_iters_max": float(iters.max()),
                        "status_nonzero": status_bad},
        }
        # executed-instruction fp64 rate of the dominant kernel (PMC: SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64
        # are per-wavefront instruction counts; x 64 lanes, FMA = 2 flops) -- hardware utilisation,
        # not the reference-algorithm FLOP model
        if all(k in kpm for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64")):
            fl = 64.0 * (2 * kpm["SQ_INSTS_VALU_FMA_F64"] + kpm["SQ_INSTS_VALU_MUL_F64"] +
                         kpm["SQ_INSTS_VALU_ADD_F64"] + kpm.get("SQ_INSTS_VALU_TRANS_F64", 0.0))
            tf = fl / (kern_ms * 1e-3) / 1e12
            out["fp64"] = {"achieved_tflops": tf, "peak_tflops": FP64_PEAK_TFLOPS, "frac": tf / FP64_PEAK_TFLOPS,
                           "basis": "executed fp64 VALU instructions of the dominant kernel (rocprofv3 PMC, all 64 "
                                    "lanes counted) / its HIP-event time",
                           "valu_insts_per_wave": kpm.get("SQ_INSTS_VALU", 0.0) / max(kpm.get("SQ_WAVES", 1.0), 1.0)}
        if elapsed2 is not None:
            out["fast_mode" if other == "per_problem" else "batch_mode"] = {
                "termination": other, "value": B_PER_GPU * args.steps / elapsed2, "unit": "QPs/sec",
                "ms_per_step": elapsed2 / args.steps * 1e3, "forward_call_ms": fwd2_ms, "backward_ms": bwd2_ms,
                "pdipm_iters_mean": float(hp2.info[:, 1].float().mean()),
                "max_abs_dzhat_vs_headline": float((hp2.zhat - zhat_head).abs().max()),
                "note": "float-tolerance parity only (include/dqp.h); every problem stops on its own"}
        if not args.no_cpu_baseline and world == 1:     # the CPU baseline is an N=1, rank-0 figure
            cb, o = cpu_baseline(host_inputs)
            out["cpu_baseline"] = cb
            err = float(np.abs(zhat_head.cpu().numpy() - o["zhat"]).max())
            out["max_abs_err_vs_cpu_zhat"] = err
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
