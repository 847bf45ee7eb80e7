"""Builds csrc/*.hip into csrc/libdqp_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(CSRC, "libdqp_hip.so")
SOURCES = ["dqp_pdipm.hip", "dqp_r16.hip"]
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
    cmd = [hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-shared", "-fPIC",
           "-fvisibility=hidden", "-o", SO] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
