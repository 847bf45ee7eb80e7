#!/usr/bin/env python3
"""Where the one-time setup of the null-space forward kernel spends its time, measured on the
production code path: one library variant per phase boundary K, in which every wavefront ends at
that boundary (-DDQP_SETUP_STOP=K: the state produced so far is kept alive by one checksum store),
timed with HIP events.  Differences between consecutive K are the phases.  (The s_memtime stamps of
tools/stamps.py perturb register allocation; these builds only remove code after the boundary.)

    python tools/setup_phases.py build      # here: compiles csrc/libdqp_hip_stop<K>.so (not the product)
    python tools/setup_phases.py            # on the GPU box: times every variant
"""
import ctypes, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_qp_mpc_amd import _build

STOPS = [(1, "A  load Q, Cholesky, Lq -> LDS"), (2, "B  G, A rows times Lq^-T"), (3, "C+D  Householder LQ, xy"),
         (4, "E+F  ph, h', w1"), (5, "G  Lq Qf -> LqZ, qv"), (15, "H  context -> workspace, Gz Gz^T")]
SIZE = os.environ.get("SIZE", "30_30_15")


def so_for(k):
    return os.path.join(_build.CSRC, "libdqp_hip_stop%d.so" % k)


def build():
    procs = []
    for k, _ in STOPS:
        objs = []
        for obj, cmd in _build._jobs():
            if os.path.basename(obj) == "dqp_r16n_%s.o" % SIZE:
                o2 = obj.replace(".o", ".stop%d.o" % k)
                procs.append(subprocess.Popen([c if c != obj else o2 for c in cmd] + ["-DDQP_SETUP_STOP=%d" % k]))
                objs.append(o2)
            else:
                objs.append(obj)
        so_for.objs = getattr(so_for, "objs", {})
        so_for.objs[k] = objs
    for pr in procs:
        assert pr.wait() == 0
    for k, _ in STOPS:
        subprocess.check_call([_build.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so_for(k)] + so_for.objs[k])
        print("built", so_for(k))


def run_one(k):
    import torch
    import bench
    from diff_qp_mpc_amd import _lib
    if isinstance(k, str):
        _build.SO = k                      # any variant library
    elif k:
        _build.SO = so_for(k)
    n, m, e = [int(t) for t in SIZE.split("_")]
    B = int(os.environ.get("BATCH", "4096"))
    hp = bench.HotPath(torch.device("cuda", 0), bench.family_R(0, B, n, m, e), termination="per_problem")
    for _ in range(3):
        hp.forward()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for _ in range(20):
        ev[0].record(); hp.forward(); ev[1].record(); torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]) * 1e3)
    ts.sort()
    print("%.2f" % ts[len(ts) // 2])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    elif len(sys.argv) > 2 and sys.argv[1] == "one":
        run_one(int(sys.argv[2]))
    elif len(sys.argv) > 2 and sys.argv[1] == "lib":
        for path in sys.argv[2:]:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "libone", path], capture_output=True, text=True)
            print("%-60s %s us" % (os.path.basename(path), out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]))
    elif len(sys.argv) > 2 and sys.argv[1] == "libone":
        run_one(os.path.abspath(sys.argv[2]))
    else:
        prev = 0.0
        rows = []
        for k, name in STOPS + [(0, "whole forward call (per-problem termination)")]:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "one", str(k)], capture_output=True, text=True)
            t = float(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 and out.stdout.strip() else float("nan")
            rows.append((name, t, t - prev))
            prev = t
        print("size %s  B=%s   (microseconds; launch overhead of an empty kernel is in the first row)" % (SIZE, os.environ.get("BATCH", "4096")))
        for name, t, d in rows:
            print("  up to %-44s %8.1f   (+%.1f)" % (name, t, d))
