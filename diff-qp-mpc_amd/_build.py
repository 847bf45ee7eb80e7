"""Builds csrc/*.hip into csrc/libdqp_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(CSRC, "libdqp_hip.so")
SOURCES = ["dqp_pdipm.hip", "dqp_r16.hip", "dqp_r16n.hip", "dqp_mpc.hip"]
ARCH = "gfx950"


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, s) for s in SOURCES]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "dqp.h"))
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    # The DPP-row kernels keep whole matrices in VGPR arrays: every loop over a register index
    # must be fully unrolled, far beyond clang's default pragma-unroll budget.
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
             "-mllvm", "-pragma-unroll-threshold=10000000", "-mllvm", "-unroll-threshold=10000000"]
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc()] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((subprocess.Popen(cmd), cmd))
        objs.append(obj)
    for pr, cmd in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    subprocess.check_call([hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", SO] + objs)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
