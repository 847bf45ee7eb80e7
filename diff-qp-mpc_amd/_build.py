"""Builds csrc/*.hip into csrc/libdqp_hip.so for gfx950 (hipcc cross-compiles without a GPU).

The DPP-row kernels keep whole matrices in VGPR arrays, so every loop over a register index is
fully unrolled (far beyond clang's default pragma-unroll budget) and each size is its own
translation unit; the objects are compiled in parallel.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(CSRC, "libdqp_hip.so")
ARCH = "gfx950"

# (nz, nineq, neq) instantiations of the 4-QPs-per-wavefront kernels (csrc/dqp_r16.hip).
# MPC shapes: nz = T(n+m), nineq = 2Tm, neq = Tn.
R16_SIZES = [
    (30, 30, 15),   # n=3 m=3 T=5: the BASELINE metric config
    (20, 10, 15),   # n=3 m=1 T=5 (PendulumDx)
    (15, 10, 10),   # n=2 m=1 T=5 (deqmpc pendulum)
    (25, 10, 20),   # n=4 m=1 T=5 (cartpole-1)
    (10, 5, 3), (12, 8, 0),   # small test sizes (with / without equalities)
]
# null-space form of the forward kernel (csrc/dqp_r16n.hip): every size above with equalities
R16N_SIZES = [s for s in R16_SIZES if s[2] > 0] + [
    # T = 10 / n = 6 MPC shapes: nz > 32 (3 register slots in setup, which spills -- it runs once),
    # but nz - neq is 10 and 5, so the iteration is tiny.  Forward + context backward only.
    (40, 20, 30),   # n=3 m=1 T=10 (BASELINE config 2: data/pendulum.pkl)
    (35, 10, 30),   # n=6 m=1 T=5  (cartpole-2)
]

PLAIN_SOURCES = ["dqp_pdipm.hip", "dqp_mpc.hip", "dqp_al.hip", "dqp_term.hip", "dqp_dyn.hip", "dqp_al_banded.hip",
                 "dqp_ric.hip", "dqp_trace.hip", "dqp_big.hip"]
SOURCES = PLAIN_SOURCES + ["dqp_r16.hip", "dqp_r16n.hip", "dqp_dispatch.hip"]
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-MD",
         "-mllvm", "-pragma-unroll-threshold=10000000", "-mllvm", "-unroll-threshold=10000000"]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def _deps():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".hpp"))]
    d.append(os.path.join(os.path.dirname(HERE), "include", "dqp.h"))
    d.append(os.path.abspath(__file__))
    return d


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(d) > t for d in _deps() if os.path.exists(d))


def _jobs():
    """-> list of (object path, hipcc argv)"""
    cc = hipcc()
    jobs = []
    variants = [("r16f", "dqp_r16.hip", R16_SIZES, []), ("r16b", "dqp_r16.hip", R16_SIZES, ["-DDQP_R16_BWD"]),
                ("r16n", "dqp_r16n.hip", R16N_SIZES, []), ("r16nb", "dqp_r16n.hip", R16N_SIZES, ["-DDQP_R16_BWD"])]
    for tag, src, sizes, extra in variants:
        for n, m, e in sorted(sizes, key=lambda t: -t[0] * t[1]):        # longest compiles first
            obj = os.path.join(CSRC, "dqp_%s_%d_%d_%d.o" % (tag, n, m, e))
            jobs.append((obj, [cc] + FLAGS + extra + ["-DDQP_R16_N=%d" % n, "-DDQP_R16_M=%d" % m,
                                                      "-DDQP_R16_E=%d" % e, "-c",
                                                      os.path.join(CSRC, src), "-o", obj]))
    for src in PLAIN_SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        jobs.append((obj, [cc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]))
    # longest compiles first (the 3-slot null-space forwards take ~4 min each, the metric-size
    # kernels ~2-3 min, everything else seconds): keeps the wall time near total CPU / cores
    def cost(j):
        name = os.path.basename(j[0])
        m = [int(t) for t in name.replace(".o", "").split("_")[-3:]] if name.count("_") >= 4 else None
        if m is None:
            return 10 ** 9 if "pdipm" in name else 1          # the generic kernels: ~2 min
        n, mm, e = m
        w = n * n * (n + mm)
        return w * (4 if "_r16n_" in name else (1 if "_r16nb_" in name else 2))
    jobs.sort(key=cost, reverse=True)
    lst = lambda sizes: " ".join("X(%d,%d,%d)" % s for s in sizes)
    obj = os.path.join(CSRC, "dqp_dispatch.o")
    jobs.append((obj, [cc] + FLAGS + ["-DDQP_R16_SIZE_LIST=" + lst(R16_SIZES),
                                      "-DDQP_R16N_SIZE_LIST=" + lst(R16N_SIZES),
                                      "-c", os.path.join(CSRC, "dqp_dispatch.hip"), "-o", obj]))
    return jobs


def _object_current(obj, cmd):
    """An object is reused when it was produced by the same command line and is newer than its
    own source and every header that source includes (the compiler's -MD dependency file; the
    per-size objects take minutes and most edits touch one file)."""
    stamp = obj + ".cmd"
    if not (os.path.exists(obj) and os.path.exists(stamp)):
        return False
    with open(stamp) as f:
        if f.read() != " ".join(cmd):
            return False
    src = cmd[cmd.index("-c") + 1]
    deps = [src]
    dfile = obj[:-2] + ".d"
    if os.path.exists(dfile):
        with open(dfile) as f:
            words = f.read().replace("\\\n", " ").split()
        deps += [w for w in words[1:] if w.startswith((CSRC, os.path.dirname(HERE)))]
    else:
        deps += [os.path.join(os.path.dirname(HERE), "include", "dqp.h")]
        deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    t = os.path.getmtime(obj)
    return all(os.path.exists(d) and os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=False, max_parallel=None):
    if not force and not needs_build():
        return SO
    jobs = _jobs()
    todo = [j for j in jobs if force or not _object_current(*j)]
    all_jobs, jobs = jobs, todo
    max_parallel = max_parallel or max(1, min(len(jobs), (os.cpu_count() or 2)))
    import time
    pending, running = list(jobs), []
    while pending or running:
        while pending and len(running) < max_parallel:
            obj, cmd = pending.pop(0)
            if verbose:
                print(" ".join(cmd))
            running.append((subprocess.Popen(cmd), cmd))
        still = []
        for pr, cmd in running:            # refill a slot as soon as ANY compile finishes
            rc = pr.poll()
            if rc is None:
                still.append((pr, cmd))
            elif rc != 0:
                for other, _ in running:
                    if other.poll() is None:
                        other.kill()
                raise subprocess.CalledProcessError(rc, cmd)
            else:
                with open(cmd[-1] + ".cmd", "w") as f:
                    f.write(" ".join(cmd))
        if len(still) == len(running):
            time.sleep(0.2)
        running = still
    subprocess.check_call([hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", SO] +
                          [obj for obj, _ in all_jobs])
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
