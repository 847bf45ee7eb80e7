#!/usr/bin/env python3
"""Benchmark of the hot path on the BASELINE.json configurations, one MI355X per rank.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {1,2,3,4,5}]

  --config 1 (default, the BASELINE metric): QPs/sec for one forward + one backward of the differentiable batched
      QP solver, batch 4096 per GPU, n_state 3, n_ctrl 3, T 5 (nz 30, nineq 30, neq 15; random dense family R of
      SURVEY.md 8d), fp64: dqp_qp_forward + dqp_qp_backward through the C ABI.  Weak scaling.
  --config 2: qp_wrapper.MPC on the PendulumDx model, batch 1024 per GPU, T 10, single-QP call, forward + backward
      (BASELINE configs[1]).  Weak scaling (replicas of the 1-GPU case).
  --config 3: AL_mpc.MPC on cartpole-1, batch 4096 per GPU, T 20, forward + backward (configs[2]).  Weak scaling.
  --config 4: AL_mpc.MPC on the rex quadrotor, GLOBAL batch 8192 sharded over the ranks, T 30 (configs[3]).  Strong.
  --config 5: one DEQ-MPC training step (deqmpc/train.py:150-175) on cartpole-2, GLOBAL batch 65536 sharded over
      the ranks, T 5, deq_iter 6, one flat gradient all-reduce (configs[4]).  Strong scaling.

A "step" is one pass of the hot path over one batch resident in HBM.  N > 1: one process per GPU.  The driver starts
them with torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment); started WITHOUT them,
`python bench.py --gpus N` starts its N ranks itself -- child processes created before this process makes any GPU
call; the parent only waits and exits non-zero if a child fails.  No rank touches another rank's QPs: the only
collectives are the single gather of the solved batch north_star names (configs 1-4) and the gradient all-reduce of
config 5.  Timing: W warm-up steps, then exactly K steps between barrier + synchronize on both sides, MAX over ranks.

Rank 0 prints ONE JSON line with `roofline` (the library kernel with the largest share of the step, timed live with
HIP events on its launch stream by the library's own trace, dqp_trace_begin / dqp_trace_end; algorithmic bytes as
DESIGN.md section 4 defines them per kernel) and `cpu_baseline` (the oracle -- a CPU port of the reference's
algorithm -- timed on this host on a bounded sample of the same workload; N = 1 only).  The default run (config 1,
one GPU) also appends `other_configs`: a short timed run of each of configs 2-5 (--no-other-configs skips them).
"""
import argparse
import ctypes
import gc
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6      # public MI355X spec (vector = matrix fp64)
PROFILE_DIR = os.path.join(ROOT, "profiles", "r3")


# ----------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without a torchrun environment
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv, env=None, python=sys.executable):
    """Start n ranks of this script as child processes and wait for them.  Called before the parent has made any GPU
    call (importing torch makes none); the children are fresh interpreters, nothing is re-exec'ed.  Returns the exit
    code: 0 iff every rank returned 0; the first failure terminates the others (by their own PIDs)."""
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                MASTER_PORT=base.get("MASTER_PORT") or str(free_port()))
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([python, os.path.abspath(__file__)] + list(argv), env=e))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.05)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in live:
                    q.terminate()
    return rc


# ----------------------------------------------------------------------------------------------------------------
def pmc_summary(config):
    """rocprofv3 --pmc summary of this command committed under profiles/r3/ (tools/profile_bench.sh): only quoted when
    it was taken on the kernel sources this run uses (fingerprint), else {}."""
    try:
        d = json.load(open(os.path.join(PROFILE_DIR, "pmc_config%d.json" % config)))
        return d if d.get("_library_fingerprint") == library_fingerprint() else {}
    except Exception:
        return {}


def library_fingerprint():
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "diff-qp-mpc_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def traffic_of(kpm):
    """HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes (KB; gfx950 FETCH_SIZE counts half of the fetched
    bytes of wide streaming reads, MI355X_MICROARCH.md HBM section)."""
    if "FETCH_SIZE" not in kpm or "WRITE_SIZE" not in kpm:
        return None
    return (2.0 * kpm["FETCH_SIZE"] + kpm["WRITE_SIZE"]) * 1024.0


def fp64_of(kpm, kern_ms):
    keys = ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64")
    if not all(k in kpm for k in keys):
        return None
    fl = 64.0 * (2 * kpm["SQ_INSTS_VALU_FMA_F64"] + kpm["SQ_INSTS_VALU_MUL_F64"] + kpm["SQ_INSTS_VALU_ADD_F64"] +
                 kpm.get("SQ_INSTS_VALU_TRANS_F64", 0.0))
    tf = fl / (kern_ms * 1e-3) / 1e12
    out = {"achieved_tflops": tf, "peak_tflops": FP64_PEAK_TFLOPS, "frac": tf / FP64_PEAK_TFLOPS,
           "basis": "executed fp64 VALU instructions of the dominant kernel (rocprofv3 PMC, all 64 lanes counted) / "
                    "its HIP-event time"}
    if kpm.get("SQ_WAVES") and kpm.get("SQ_INSTS_VALU"):
        out["valu_insts_per_wave"] = kpm["SQ_INSTS_VALU"] / kpm["SQ_WAVES"]
        out["fp64_share_of_valu"] = (kpm["SQ_INSTS_VALU_FMA_F64"] + kpm["SQ_INSTS_VALU_MUL_F64"] +
                                     kpm["SQ_INSTS_VALU_ADD_F64"] + kpm.get("SQ_INSTS_VALU_TRANS_F64", 0.0)) / kpm["SQ_INSTS_VALU"]
    return out


def short_kernel(name):
    """'void dqp::r16n::forward_kernel<dqp::r16n::Cfg<30, 30, 15> >(dqp::KParams)' -> 'r16n::forward_kernel<Cfg<30,30,15>>'
    (the HIP runtime's and rocprofv3's spellings of a kernel name reduce to the same string)"""
    import re
    n = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("dqp::dyn::", "").replace("dqp::", "")
    if "(" in n:
        n = n[:n.rindex("(")]
    n = re.sub(r"\b(r16n|r16|ric)::(?=Cfg|RES_)", "", n)
    return n.replace(" >", ">").replace(", ", ",").strip()


def mpc_sizes(n, m, T):
    nt = n + m
    return dict(nt=nt, nz=T * nt, neq=T * n, nineq=2 * T * m)


def mpc_qp_forward_bytes(n, m, T):
    """dqp_mpc_qp_forward: C, c, F, f, x0 in; tau, lam, nu, slack out (doubles x 8)"""
    z = mpc_sizes(n, m, T)
    nt = z["nt"]
    return 8 * (T * nt * nt + T * nt + (T - 1) * n * nt + (T - 1) * n + n + z["nz"] + 2 * z["nineq"] + z["neq"])


def mpc_qp_backward_bytes(n, m, T):
    """dqp_mpc_qp_backward: C, F, tau, lam, nu, slack, dl/dtau in; dC, dc, dF, df, dx0 out"""
    z = mpc_sizes(n, m, T)
    nt = z["nt"]
    inp = T * nt * nt + (T - 1) * n * nt + 2 * z["nz"] + 2 * z["nineq"] + z["neq"]
    return 8 * (inp + T * nt * nt + T * nt + (T - 1) * n * nt + (T - 1) * n + n)


def al_newton_bytes(n, m, T):
    """al_banded_newton_kernel: xu, Qdiag, q, lam, rho, x0 in; the Newton update and the block-tridiagonal factor
    (nt rows x (nt + 1 + n) per knot: what the backward sweep and NewtonAL.backward read) out"""
    nt = n + m
    return 8 * (3 * T * nt + T * n + 2 * T * m + 1 + n + T * nt + T * nt * (nt + 1 + n))


def al_ls_bytes(n, m, T):
    """al_ls_group_kernel: xu, update, Qdiag, q, lam, rho, x0 in; 20 merit values out"""
    nt = n + m
    return 8 * (4 * T * nt + T * n + 2 * T * m + 1 + n + 20)


# ----------------------------------------------------------------------------------------------------------------
# workloads
class Workload:
    config = 0
    scaling = "weak"
    unit = "QPs/sec"

    def gathered_source(self):
        """the solved batch this rank contributes to the single gather (a contiguous fp64 tensor), or None"""
        return None

    def kernel_bytes(self, kernel):
        """algorithmic HBM bytes of one launch of `kernel` (short name) on this rank's batch, or None"""
        return None

    def extras(self):
        return {}


def family_R(torch, seed, B, nz, nineq, neq):
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


class MetricQP(Workload):
    """config 1: the C-ABI calls themselves on pre-allocated buffers."""
    config = 1
    NZ, NINEQ, NEQ = 30, 30, 15
    B_PER_GPU = 4096
    metric = "QPs/sec (fwd+bwd), batch=4096 n=3 m=3 T=5"
    # algorithmic elements per QP
    FWD_ELEMS = (30 * 30 + 30 + 30 * 30 + 30 + 15 * 30 + 15) + (30 + 2 * 30 + 15)                       # 2325 + 105
    BWD_ELEMS = (30 * 30 + 30 * 30 + 15 * 30 + 30 + 2 * 30 + 15 + 30) + (30 * 30 + 30 + 30 * 30 + 30 + 15 * 30 + 15)  # 2385 + 2325

    def __init__(self, torch, dev, rank, world, args, termination=None):
        from diff_qp_mpc_amd import _lib
        self.torch, self._lib, self.lib, self.dev = torch, _lib, _lib.load(), dev
        NZ, NINEQ, NEQ = self.NZ, self.NINEQ, self.NEQ
        self.termination = termination or args.termination
        B = self.B = self.B_PER_GPU
        self.units = B
        self.host_inputs = family_R(torch, rank, B, NZ, NINEQ, NEQ)
        self.Q, self.p, self.G, self.h, self.A, self.b = [t.to(dev).contiguous() for t in self.host_inputs]
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
        # "batch": the reference's batch-coupled stopping rule replayed on the device (parity-safe, the package
        # default); "per_problem": every QP stops on its own (include/dqp.h)
        tflag = _lib.DQP_FLAG_BATCH_TERMINATION if self.termination == "batch" else 0
        self.opts = _lib.dqp_opts(float(os.environ.get("DQP_BENCH_EPS", "1e-12")), 1e-10, 20, 3, tflag, 0)
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
        self.bopts = _lib.dqp_opts(0.0, 0.0, 0, 0, _lib.DQP_FLAG_BACKWARD_CTX if wsb > 0 else 0, 0)

    def forward(self):
        rc = self.lib.dqp_qp_forward(ctypes.byref(self.dims), ctypes.byref(self.opts), *self.fargs,
                                     self.wsp, self.termp, self.stream)
        if rc:
            raise RuntimeError("dqp_qp_forward rc=%d" % rc)

    def backward(self):
        rc = self.lib.dqp_qp_backward(ctypes.byref(self.dims), ctypes.byref(self.bopts), *self.bargs,
                                      self.null, self.wsp, self.stream)
        if rc:
            raise RuntimeError("dqp_qp_backward rc=%d" % rc)

    def step(self, gather=None):
        self.forward()
        work = gather(self.zhat) if gather else None          # overlaps the backward kernel
        self.backward()
        if work is not None:
            work.wait()

    def gathered_source(self):
        return self.zhat

    def kernel_bytes(self, kernel):
        if "r16n::forward_kernel" in kernel:
            return self.FWD_ELEMS * 8 * self.B
        if "backward_kernel" in kernel:
            return self.BWD_ELEMS * 8 * self.B
        return None

    def describe(self, world):
        return {"workload": "random dense QP family R (SURVEY 8d), configs[0] shape at the BASELINE metric batch: "
                            "B=4096/GPU nz=30 nineq=30 neq=15",
                "global_batch": world * self.B, "n_state": 3, "n_ctrl": 3, "T": 5,
                "parallelism": "batch-shard x%d" % world,
                "termination": self.termination + (" (the reference's batch-coupled rule, parity-safe)"
                                                   if self.termination == "batch" else " (per-problem exit)")}

    def extras(self):
        iters = self.info[:, 1].float()
        return {"pdipm_iters_mean": float(iters.mean()), "pdipm_iters_max": float(iters.max()),
                "status_nonzero": int((self.info[:, 0] != 0).sum())}

    def cpu_baseline(self, reps=3):
        """The oracle (a C port of the reference's algorithm, OpenMP over the batch) on this host."""
        import numpy as np
        from oracle import oracle
        Q, p, G, h, A, b = [t.numpy() for t in self.host_inputs]
        B = Q.shape[0]
        nthreads = oracle.max_threads()
        ct = np.ones((B, self.NZ))
        ts = []
        for r in range(reps + 1):
            t0 = time.perf_counter()
            o = oracle.qp_forward(Q, p, G, h, A, b, nthreads=nthreads)
            oracle.qp_backward(Q, G, A, o["zhat"], o["lam"], o["nu"], o["slack"], ct, nthreads=nthreads)
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts[1:]))
        err = float(np.abs(self.zhat.cpu().numpy() - o["zhat"]).max())
        return {"value": B / t, "unit": "QPs/sec", "cores": nthreads, "kind": "port",
                "sample": "same workload, %d QPs fwd+bwd, median of %d reps after 1 warm-up; the batch-coupled PDIPM "
                          "ran %d iterations on every QP, as the GPU headline does" % (B, reps, o["iters"]),
                "max_abs_err_gpu_vs_cpu_zhat": err}


class PendulumMPC(Workload):
    """config 2: qp_wrapper.MPC on the PendulumDx device model (n 3, m 1), T 10, single-QP call."""
    config = 2
    unit = "MPC solves/sec"
    metric = "MPC QP solves/sec (qp_wrapper.MPC fwd+bwd), PendulumDx batch=1024 T=10"
    n, m, T, B_PER_GPU = 3, 1, 10, 1024

    def __init__(self, torch, dev, rank, world, args):
        import numpy as np
        from diff_qp_mpc_amd import qp_wrapper
        from diff_qp_mpc_amd.dynamics import DeviceDynamics
        self.torch, self.qp_wrapper = torch, qp_wrapper
        B = self.B = self.units = self.B_PER_GPU
        T, n, m = self.T, self.n, self.m
        self.dyn = DeviceDynamics("pendulum_dx")
        rng = np.random.default_rng(rank)
        th = rng.uniform(-np.pi / 2, np.pi / 2, B)          # il_env_nonconvex.py:62-65
        f64 = dict(dtype=torch.float64, device=dev)
        self.x0 = torch.tensor(np.stack([np.cos(th), np.sin(th), rng.uniform(-1, 1, B)], 1), **f64)
        goal = torch.tensor([1.0, 0.0, 0.0, 0.0], **f64)
        Qw = torch.tensor([1.0, 1.0, 0.1, 0.001], **f64)
        self.C = torch.diag(Qw).repeat(T, B, 1, 1).requires_grad_()
        self.c = (-(Qw * goal)).repeat(T, B, 1).requires_grad_()
        lo, hi = torch.tensor([-2.0], **f64), torch.tensor([2.0], **f64)
        self.mpc = qp_wrapper.MPC(n, m, T, u_lower=lo, u_upper=hi, n_batch=B, verbose=-1, single_qp_solve=True)
        self.u = None
        self.graphed = None
        if getattr(args, "graph", False):      # the call and its backward as two hipGraphs (qp_wrapper.GraphedMPC)
            self.graphed = qp_wrapper.graphed_mpc(self.mpc, (self.x0, self.C, self.c), lambda: (self.dyn, self.dyn.jac))

    def step(self, gather=None):
        if self.graphed is not None:
            x, u = self.graphed(self.x0, self.C, self.c)
        else:
            x, u = self.mpc(self.x0, self.qp_wrapper.QuadCost(self.C, self.c), self.dyn, self.dyn.jac)
        self.u = u.detach().transpose(0, 1).contiguous()
        work = gather(self.u) if gather else None
        self.C.grad = self.c.grad = None
        (x.sum() + u.sum()).backward()
        if work is not None:
            work.wait()

    def gathered_source(self):
        return self.u

    def kernel_bytes(self, kernel):
        if "ric::forward_kernel" in kernel:
            return mpc_qp_forward_bytes(self.n, self.m, self.T) * self.B
        if "ric::backward_kernel" in kernel:
            return mpc_qp_backward_bytes(self.n, self.m, self.T) * self.B
        return None

    def describe(self, world):
        return {"workload": "BASELINE configs[1]: qp_wrapper.MPC, PendulumDx (n 3, m 1), T 10, single-QP call, x0 as "
                            "il_env_nonconvex.py:62-65; stage-wise PDIPM with the true-dynamics residual + line search + "
                            "backward; B=1024/GPU" + (", replayed as hipGraphs" if self.graphed else ""),
                "global_batch": world * self.B, "n_state": 3, "n_ctrl": 1, "T": 10,
                "parallelism": "replicas x%d" % world}

    def extras(self):
        return {"max_abs_u": float(self.u.abs().max())}

    def cpu_baseline(self):
        """DenseQPFunction (batch_LU.py) forward + backward of the assembled QP of this call, C oracle, OpenMP."""
        import numpy as np
        from oracle import oracle
        torch, qw = self.torch, self.qp_wrapper
        with torch.no_grad():
            u0 = torch.zeros(self.T, self.B, self.m, dtype=torch.float64, device=self.x0.device)
            x = self.mpc.rollout(self.x0, u0, self.dyn)
            F, f = self.mpc.linearize_dynamics(x, u0, self.dyn, self.dyn.jac, diff=False)
            Q, p, G, h, A, b = self.mpc._dense(self.C.detach(), self.c.detach(), F, f, self.x0)
        host = [t.cpu().numpy() for t in (Q, p, G, h, A, b)]
        B = host[0].shape[0]
        nthreads = oracle.max_threads()
        ct = np.ones((B, host[0].shape[1]))
        ts = []
        for r in range(3):
            t0 = time.perf_counter()
            o = oracle.dense_forward(*host, nthreads=nthreads)
            oracle.dense_backward(o["K"], o["zhat"], o["lam"], o["nu"], ct, nthreads=nthreads)
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts[1:]))
        return {"value": B / t, "unit": self.unit, "cores": nthreads, "kind": "port",
                "sample": "the dense QP of the first linearisation of the same %d problems (nz 40, nineq 20, neq 30): "
                          "DenseQPFunction forward + backward as qp_wrapper.MPC.single_qp runs it (110 x 110 KKT LU per "
                          "iteration), without linearisation, assembly and line search; median of 2 reps after 1 warm-up" % B}


class RobotAL(Workload):
    """configs 3 and 4: AL_mpc.MPC on a registered robot (2 AL iterations x 4 Newton steps + backward)."""
    unit = "trajectories/sec"

    def __init__(self, torch, dev, rank, world, args, robot, B, T):
        import numpy as np
        from diff_qp_mpc_amd import AL_mpc, al_utils
        from diff_qp_mpc_amd.dynamics import DeviceDynamics
        self.torch, self.al_utils, self.robot = torch, al_utils, robot
        self.B = self.units = B
        self.T = T
        self.dyn = dyn = DeviceDynamics(robot)
        nx, nu = self.nx, self.nu = dyn.n_state, dyn.n_ctrl
        rng = np.random.default_rng(rank)
        f64 = dict(dtype=torch.float64, device=dev)
        if robot == "rexquadrotor":         # setup as tests/golden/make_golden_cfg4.py
            win = np.array([1.0] * 3 + [0.15] * 3 + [0.5] * 3 + [0.25] * 3)
            x0 = torch.tensor(rng.uniform(-1, 1, (B, nx)) * win, **f64)
            Qw = torch.tensor([10.0] * 3 + [0.01] * 3 + [1.0] * 3 + [0.01] * 3 + [1e-4] * nu, **f64)
            lo, hi = torch.full((nu,), 11.5, **f64), torch.full((nu,), 18.3, **f64)
            u_ref = torch.full((B, T, nu), (2.0 * 9.81 + 4 * 30.48576) / (4 * 0.0244101 * 100.0), **f64)
        else:                               # tests/golden/make_golden_cfg3.py (cartpole.py:66-78,121-131)
            x0 = torch.tensor(rng.uniform(-np.pi, np.pi, (B, nx)), **f64)
            Qw = torch.cat([torch.ones(nx), 1e-8 * torch.ones(nu)]).to(**f64)
            ub = 100.0 if robot == "cartpole1l" else 250.0
            lo, hi = torch.full((nu,), -ub, **f64), torch.full((nu,), ub, **f64)
            u_ref = torch.zeros(B, T, nu, **f64)
        self.x0, self.lo, self.hi = x0, lo, hi
        self.Qd = Qw.repeat(B, T, 1)
        self.x_ref = x0[:, None, :] * torch.linspace(1.0, 0.0, T, **f64)[None, :, None]
        self.u_ref = u_ref
        self.C = torch.diag_embed(self.Qd).requires_grad_()
        self.c = (-(self.Qd * torch.cat([self.x_ref, u_ref], -1))).clone().requires_grad_()
        self.ctrl = AL_mpc.MPC(nx, nu, T, u_lower=lo, u_upper=hi, n_batch=B, verbose=0, solver_type="dense",
                               dtype=torch.float64, eps=1e-5, exit_unconverged=False, backprop=False)
        self.mask = torch.ones(B, T, 1, device=dev)
        self.ctrl.mask = self.mask
        self.u = None
        self.graphed = None
        if getattr(args, "graph", False):      # the cold call captured as two hipGraphs (AL_mpc.GraphedMPC)
            self.graphed = AL_mpc.GraphedMPC(self.ctrl, (self.x0, self.C, self.c), dyn, x_init=self.x_ref, u_init=self.u_ref)

    def step(self, gather=None):
        if self.graphed is not None:
            x, u = self.graphed(self.x0, self.C, self.c)
        else:
            self.ctrl.reinitialize(self.x0, self.mask)
            self.ctrl.x_init, self.ctrl.u_init = self.x_ref, self.u_ref
            x, u = self.ctrl(self.x0, self.al_utils.QuadCost(self.C, self.c), self.dyn, self.dyn.jac)
        self.x, self.u = x.detach(), u.detach().double().contiguous()
        work = gather(self.u) if gather else None
        self.C.grad = self.c.grad = None
        (x.double().sum() + 2.0 * u.double().sum()).backward()
        if work is not None:
            work.wait()

    def gathered_source(self):
        return self.u

    def kernel_bytes(self, kernel):
        if "al_banded_newton_kernel" in kernel:
            return al_newton_bytes(self.nx, self.nu, self.T) * self.B
        if "al_ls_group_kernel" in kernel:
            return al_ls_bytes(self.nx, self.nu, self.T) * self.B
        return None

    def extras(self):
        torch = self.torch
        xs, us = self.x.double(), self.u
        gap = (self.dyn(xs[:, :-1].reshape(-1, self.nx), us[:, :-1].reshape(-1, self.nu)).reshape(self.B, self.T - 1, self.nx)
               - xs[:, 1:]).abs().amax(dim=(1, 2))
        return {"dynamics_gap_median": float(gap.median()), "dynamics_gap_max": float(gap.max()),
                "finite": bool(torch.isfinite(xs).all())}

    def cpu_baseline(self, budget_s=12.0):
        """oracle/al_solve_oracle.py (numpy port of AL_mpc.MPC.al_solve + NewtonAL.backward), one thread."""
        return al_cpu_baseline(self.robot, self.dyn.dt, self.x0, self.Qd, self.c.detach(), self.lo, self.hi, self.x_ref,
                               self.u_ref, calls=1, budget_s=budget_s, unit=self.unit)


def al_cpu_baseline(robot, dt, x0, Qd, c, lo, hi, x_ref, u_ref, calls, budget_s, unit):
    import numpy as np
    from oracle import al_solve_oracle as aso, dyn_host
    try:
        from threadpoolctl import threadpool_limits
    except Exception:           # pragma: no cover
        threadpool_limits = None
    step = dyn_host.stepper(robot, dt)
    if step is None:
        return None
    h = lambda t: t.detach().cpu().numpy().astype(np.float64)
    x0, Qd, c, lo, hi, xr, ur = [h(t) for t in (x0, Qd, c, lo, hi, x_ref, u_ref)]
    B, T, nt = Qd.shape
    n = x0.shape[1]
    m = nt - n

    def run(S):
        lam, rho, hist = np.zeros((S, T * n + 2 * T * m)), np.ones((S, 1)), None
        x, u = xr[:S], ur[:S]
        g = np.concatenate((np.ones((S, T, n)), 2.0 * np.ones((S, T, m))), 2)
        t0 = time.perf_counter()
        for _ in range(calls):          # config 5: deq_iter successive solves, warm-started like the policy's
            o = aso.al_solve(x, u, x0[:S], Qd[:S], c[:S], lo, hi, step, lam, rho, history=hist)
            aso.backward(o["L"], o["xu"], g)
            x, u, lam, rho, hist = o["x"], o["u"], o["lam"], o["rho"], o["history"]
        return time.perf_counter() - t0

    import contextlib
    with (threadpool_limits(limits=1) if threadpool_limits else contextlib.nullcontext()):
        probe = min(8, B)
        t_probe = run(probe)
        S = int(max(probe, min(B, budget_s / (t_probe / probe))))
        t = run(S)
    return {"value": S / t, "unit": unit, "cores": 1, "kind": "port",
            "sample": "%d of the same trajectories x %d AL_mpc.MPC call(s) (2 AL iterations x 4 Newton steps with dense "
                      "Jacobians, %d x %d Hessian Cholesky, 20-candidate line search, as the reference) + NewtonAL.backward; "
                      "numpy oracle (oracle/al_solve_oracle.py), BLAS pinned to one thread, %.1f s" % (S, calls, T * nt, T * nt, t)}


class CartpoleAL(RobotAL):
    config = 3
    metric = "trajectories/sec (AL_mpc.MPC fwd+bwd), cartpole-1 batch=4096 T=20"

    def __init__(self, torch, dev, rank, world, args):
        # --robot / --T / --batch: the same call at other sizes (the reference trains at --bsz 128, deqmpc/train.py:46)
        robot = getattr(args, "robot", None) or "cartpole1l"
        super().__init__(torch, dev, rank, world, args, robot, getattr(args, "batch", None) or 4096, getattr(args, "T", None) or 20)
        if (robot, self.B, self.T) != ("cartpole1l", 4096, 20):
            self.metric = "trajectories/sec (AL_mpc.MPC fwd+bwd), %s batch=%d T=%d" % (robot, self.B, self.T)

    def describe(self, world):
        std = (self.robot, self.B, self.T) == ("cartpole1l", 4096, 20)
        return {"workload": ("BASELINE configs[2]: " if std else "variant of configs[2]: ") +
                            "AL_mpc.MPC inner loop on %s (n %d, m %d), T %d, 2 AL iterations x 4 Newton steps + backward, setup as "
                            "tests/golden/make_golden_cfg3.py; B=%d/GPU%s" % (self.robot, self.nx, self.nu, self.T, self.B,
                                                                             ", cold call replayed as hipGraphs" if self.graphed else ""),
                "global_batch": world * self.B, "n_state": self.nx, "n_ctrl": self.nu, "T": self.T,
                "parallelism": "batch-shard x%d" % world}


class QuadrotorAL(RobotAL):
    config = 4
    scaling = "strong"
    metric = "trajectories/sec (AL_mpc.MPC fwd+bwd), rex_quadrotor batch=8192 T=30"
    GLOBAL_B = 8192

    def __init__(self, torch, dev, rank, world, args):
        assert self.GLOBAL_B % world == 0
        super().__init__(torch, dev, rank, world, args, "rexquadrotor", self.GLOBAL_B // world, 30)

    def describe(self, world):
        return {"workload": "BASELINE configs[3]: AL_mpc.MPC on the rex quadrotor (the reference's env has n_state 12, m 4; nz 480), "
                            "T 30, 2 AL iterations x 4 Newton steps + backward, setup as tests/golden/make_golden_cfg4.py; "
                            "global batch 8192 sharded", "global_batch": self.GLOBAL_B, "n_state": 12, "n_ctrl": 4, "T": 30,
                "parallelism": "batch-shard x%d (%d per GPU)" % (world, self.B)}


class DEQMPCTrain(Workload):
    """config 5: one DEQ-MPC imitation-learning step (deqmpc/train.py:150-175) on cartpole-2."""
    config = 5
    scaling = "strong"
    unit = "trajectories/sec"
    metric = "trajectories/sec (DEQ-MPC training step), cartpole-2 batch=65536 T=5 deq_iter=6"
    GLOBAL_B = 65536

    def __init__(self, torch, dev, rank, world, args):
        import types
        import numpy as np
        import torch.distributed as dist
        from diff_qp_mpc_amd import policies
        from diff_qp_mpc_amd.dynamics import DeviceDynamics
        assert self.GLOBAL_B % world == 0
        self.torch, self.policies = torch, policies
        B = self.B = self.units = int(getattr(args, "batch", None) or os.environ.get("DQP_BENCH_CFG5_BATCH", self.GLOBAL_B)) // world
        T, self.deq_iter = 5, 6
        self.T = T
        self.dyn = dyn = DeviceDynamics("cartpole2l", dt=0.03)
        nx, nu = self.nx, self.nu = dyn.n_state, dyn.n_ctrl
        env = types.SimpleNamespace(nx=nx, nu=nu, nq=nx // 2, dt=dyn.dt, dynamics=dyn, dynamics_derivatives=dyn.jac,
                                    action_space=types.SimpleNamespace(high=np.array([250.0] * nu), low=np.array([-250.0] * nu)))
        a = argparse.Namespace(T=T, nq=nx // 2, hdim=128, layer_type="mlp", deq_out_type=1, policy_out_type=1,
                               deq_iter=self.deq_iter, solver_type="al", qp_iter=1, eps=1e-2, warm_start=True, bsz=B,
                               Q=torch.ones(nx), R=1e-2 * torch.ones(nu), dtype="double", device=str(dev))
        torch.manual_seed(0)
        self.policy = policies.DEQMPCPolicy(a, env)
        self.graph = bool(getattr(args, "graph", False))
        self.opt = torch.optim.Adam(self.policy.model.parameters(), lr=1e-4, capturable=self.graph)
        gen = torch.Generator(device=dev).manual_seed(rank)
        self.x = torch.rand(B, nx, device=dev, generator=gen) - 0.5
        self.gs = self.x[:, None, :] * torch.linspace(1, 0, T, device=dev)[None, :, None]
        self.ga = torch.zeros(B, T, nu, device=dev)
        self.mask = torch.ones(B, T, device=dev)
        self.group = dist.group.WORLD if world > 1 else None
        self.loss = None
        self.graphed = None
        if self.graph:      # the whole training step as one hipGraph (policies.GraphedTrainStep)
            self.graphed = policies.GraphedTrainStep(self.policy, self.opt, self.x, self.gs, self.ga, self.mask, group=self.group)
        if B * world != self.GLOBAL_B:
            self.metric = "trajectories/sec (DEQ-MPC training step), cartpole-2 batch=%d T=5 deq_iter=6" % (B * world)

    def step(self, gather=None):
        if self.graphed is not None:
            self.loss, _, _ = self.graphed(self.x, self.gs, self.ga, self.mask)
            return
        self.loss, _, _ = self.policies.train_step(self.policy, self.opt, self.x, self.gs, self.ga, self.mask, group=self.group)

    def kernel_bytes(self, kernel):
        if "al_banded_newton_kernel" in kernel:
            return al_newton_bytes(self.nx, self.nu, self.T) * self.B
        if "al_ls_group_kernel" in kernel:
            return al_ls_bytes(self.nx, self.nu, self.T) * self.B
        return None

    def describe(self, world):
        return {"workload": "BASELINE configs[4]: DEQ-MPC training step (deqmpc/train.py:150-175 with its defaults T 5, deq_iter 6, "
                            "hdim 128, solver 'al') on cartpole-2 (n 6, m 1): 6 x [DEQLayer -> AL_mpc.MPC], L1 loss on every "
                            "iterate, backward through the solvers, one flat gradient all-reduce, Adam step; global batch "
                            "%d sharded" % (self.B * world), "global_batch": self.B * world, "n_state": 6, "n_ctrl": 1,
                "T": 5, "deq_iter": 6, "parallelism": "data-parallel x%d (%d per GPU), 1 gradient all-reduce per step" % (world, self.B)}

    def extras(self):
        return {"loss": float(self.loss)}

    def cpu_baseline(self, budget_s=12.0):
        torch = self.torch
        f64 = dict(dtype=torch.float64, device=self.x.device)
        Qd = torch.cat([torch.ones(self.nx, **f64), 1e-2 * torch.ones(self.nu, **f64)]).repeat(self.B, self.T, 1)
        xu_ref = torch.cat([self.gs.double(), self.ga.double()], -1)
        out = al_cpu_baseline("cartpole2l", self.dyn.dt, self.x.double(), Qd, -(Qd * xu_ref),
                              torch.full((self.nu,), -250.0, **f64), torch.full((self.nu,), 250.0, **f64),
                              self.gs.double(), self.ga.double(), calls=self.deq_iter, budget_s=budget_s, unit=self.unit)
        if out:
            out["sample"] += "; the six solver calls of a training step, tracking the ground-truth trajectory (the DEQLayer "\
                             "MLP and the optimiser are not in the CPU figure)"
        return out


WORKLOADS = {1: MetricQP, 2: PendulumMPC, 3: CartpoleAL, 4: QuadrotorAL, 5: DEQMPCTrain}
DEFAULT_STEPS = {1: (200, 10), 2: (50, 5), 3: (30, 3), 4: (10, 2), 5: (5, 2)}


# ----------------------------------------------------------------------------------------------------------------
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", type=int, default=1, choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default run (config 1, one GPU): do not append the short runs of configs 2-5 (`other_configs`)")
    ap.add_argument("--batch", type=int, default=None, help="configs 3 and 5: another batch size (per GPU for 3, global for 5)")
    ap.add_argument("--robot", default=None, help="config 3: another registered model (cartpole2l, pendulum_euler, ...)")
    ap.add_argument("--T", type=int, default=None, help="config 3: another horizon")
    ap.add_argument("--graph", action="store_true", help="configs 3-5: replay the call / the training step as hipGraphs")
    ap.add_argument("--termination", choices=["batch", "per_problem"], default="batch",
                    help="config 1: mode of the headline number (default: the parity-safe batch rule)")
    args = ap.parse_args(argv)
    if args.steps is None:
        args.steps = DEFAULT_STEPS[args.config][0]
    if args.warmup is None:
        args.warmup = DEFAULT_STEPS[args.config][1]

    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:       # no launcher around us: start the ranks ourselves, before any GPU call
            return launch_ranks(args.gpus, argv)
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, os.environ["WORLD_SIZE"]),
              file=sys.stderr)
        return 2

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("DQP_BENCH_DRY_RUN") == "1":
        # launcher rehearsal without a GPU (tests/test_bench_launcher.py): the ranks rendezvous over gloo, rank 0
        # prints the line's launcher-dependent fields
        seen = 1
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
            t = torch.ones(1)
            dist.all_reduce(t)
            seen = int(t.item())
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_seen": seen, "config": args.config,
                              "steps": args.steps, "warmup": args.warmup, "scaling": WORKLOADS[args.config].scaling}))
        return int(os.environ.get("DQP_BENCH_DRY_RUN_FAIL_RANK", "-1")) == rank
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

    from diff_qp_mpc_amd import _lib
    if os.environ.get("DQP_BENCH_AL_PER_SOLVE_CALLS") == "1":       # A/B: round 2's host loop around the Newton solves
        from diff_qp_mpc_amd import AL_mpc
        AL_mpc.ONE_CALL_SOLVE = False
    wl = WORKLOADS[args.config](torch, dev, rank, world, args)

    gathered = {}

    def gather(src):            # north_star: a single RCCL gather of the solved batch, overlapped with backward
        if world == 1:
            return None
        key = tuple(src.shape)
        if key not in gathered:
            gathered[key] = torch.empty((world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=dev)
        return dist.all_gather_into_tensor(gathered[key], src, async_op=True)

    for _ in range(args.warmup):
        wl.step(gather)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    gc.collect(); gc.disable()          # no collector pauses inside the timed region (as timeit)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step(gather)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel times of the same step, HIP events on the launch stream (the library's own trace)
    tsteps = max(1, min(args.steps, 20))
    graphed, wl.graphed = getattr(wl, "graphed", None), None       # graph replays launch nothing through the library: trace the eager step
    with _lib.trace(20000) as tr:          # (a step is at most ~150 library launches; the event pool is per slot)
        for _ in range(tsteps):
            wl.step(None)
        torch.cuda.synchronize()
    wl.graphed = graphed
    kern = {short_kernel(k): (c / tsteps, ms) for k, (c, ms) in tr.by_kernel().items()}
    total_kernel_ms = sum(c * ms for c, ms in kern.values())

    out = None
    if rank == 0:
        value = world * wl.units * args.steps / elapsed
        dom = max(kern, key=lambda k: kern[k][0] * kern[k][1])
        dom_launches, dom_ms = kern[dom]
        dom_bytes = wl.kernel_bytes(dom)
        pm = pmc_summary(args.config)
        kpm = next((v for k, v in pm.items() if isinstance(v, dict) and short_kernel(k) == dom), {})
        gbs = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_bytes else None
        out = {
            "metric": wl.metric, "value": value, "unit": wl.unit, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": wl.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic", "config": dict(wl.describe(world), baseline_config=args.config),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS if gbs else None, "traffic": traffic_of(kpm),
                         "traffic_source": ("profiles/r3/pmc_config%d.json (rocprofv3 --pmc passes of this command on "
                                            "these kernel sources)" % args.config if traffic_of(kpm) is not None else None),
                         "avg_launch_ms": dom_ms, "launches_per_step": dom_launches,
                         "share_of_step_kernel_time": dom_launches * dom_ms / total_kernel_ms if total_kernel_ms else None,
                         "algorithmic_bytes_per_launch": dom_bytes,
                         "note": "dominant library kernel of the step by total time, HIP events on the launch stream "
                                 "(dqp_trace_*); these kernels are fp64-issue / latency bound, not HBM bound (DESIGN.md "
                                 "section 4) -- see `fp64` for the executed-instruction rate"},
            "kernels": {k: {"launches_per_step": round(c, 2), "avg_ms": ms,
                            "GBps": (wl.kernel_bytes(k) / (ms * 1e-3) / 1e9 if wl.kernel_bytes(k) else None)}
                        for k, (c, ms) in sorted(kern.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:8]},
            "kernel_ms_per_step": total_kernel_ms,
        }
        out["kernels_extra"] = wl.extras()
        f64 = fp64_of(kpm, dom_ms) if kpm else None
        if f64:
            out["fp64"] = f64

    if args.config == 1 and world == 1:       # the other termination mode, same inputs
        other = "per_problem" if wl.termination == "batch" else "batch"
        zhat_head = wl.zhat.clone()
        wl2 = MetricQP(torch, dev, rank, world, args, termination=other)
        for _ in range(args.warmup):
            wl2.step(None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            wl2.step(None)
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t0
        out["fast_mode" if other == "per_problem" else "batch_mode"] = {
            "termination": other, "value": wl2.units * args.steps / e2, "unit": "QPs/sec",
            "ms_per_step": e2 / args.steps * 1e3, "pdipm_iters_mean": float(wl2.info[:, 1].float().mean()),
            "max_abs_dzhat_vs_headline": float((wl2.zhat - zhat_head).abs().max()),
            "note": "float-tolerance parity only (include/dqp.h); every problem stops on its own"}

    if args.config == 1 and world == 1 and not args.no_other_configs and args.batch is None:
        # the other BASELINE configurations, each a short timed run of the same kind (warm-up, K steps between
        # synchronisations, the library's trace for the dominant kernel): in the default line so that whoever runs
        # `python bench.py` has all five on record; `--config C` gives a configuration its full line
        out["other_configs"] = {}
        for c in (2, 3, 4, 5):
            try:
                k_steps, k_warm = DEFAULT_STEPS[c]
                k_warm = max(k_warm, 5)
                w = WORKLOADS[c](torch, dev, rank, world, args)
                for _ in range(k_warm):
                    w.step(None)
                torch.cuda.synchronize()
                gc.collect(); gc.disable()      # (as timeit does: a generation-2 collection over this process's heap -- the
                t0 = time.perf_counter()        # trace records of the runs before -- is a 45 ms pause in a 1.5 ms step)
                for _ in range(k_steps):
                    w.step(None)
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
                gc.enable()
                with _lib.trace(2000) as tr:
                    w.step(None)
                    torch.cuda.synchronize()
                kk = {short_kernel(k): (cnt, ms) for k, (cnt, ms) in tr.by_kernel().items()}
                dom = max(kk, key=lambda k: kk[k][0] * kk[k][1])
                out["other_configs"][str(c)] = {
                    "metric": w.metric, "value": w.units * k_steps / el, "unit": w.unit, "steps": k_steps, "warmup": k_warm,
                    "ms_per_step": el / k_steps * 1e3, "scaling": w.scaling, "dominant_kernel": dom,
                    "dominant_kernel_launches_per_step": kk[dom][0], "dominant_kernel_avg_ms": kk[dom][1],
                    "kernel_ms_per_step": sum(cnt * ms for cnt, ms in kk.values())}
                del w
                torch.cuda.empty_cache()
            except Exception as e:            # a failing side run must not cost the headline line
                out["other_configs"][str(c)] = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0:
        if not args.no_cpu_baseline and world == 1:     # the CPU baseline is an N=1, rank-0 figure
            cb = wl.cpu_baseline()
            if cb:
                out["cpu_baseline"] = cb
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
