#!/usr/bin/env python3
"""A/B builds of one kernel source: links csrc/libdqp_hip_ab_<name>.so with the object of SOURCE
replaced by a compile of each variant file (and extra -D flags), so that several versions can be timed
in ONE gpurun call (box-to-box differences are ~10 %, larger than most single changes).

    python tools/ab_variants.py build dqp_ric.hip name1=/path/to/variant1.hip[,-DX=1] name2=...
    DQP_HIP_LIBRARY=diff-qp-mpc_amd/csrc/libdqp_hip_ab_name1.so python tools/bench_ric.py
"""
import os, shutil, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diff_qp_mpc_amd import _build


def build(source, specs):
    jobs = list(_build._jobs())
    # SOURCE is a .hip file with one object, or the object itself for the per-size translation units
    # (dqp_r16n_30_30_15.o: the metric-size null-space forward, compiled from dqp_r16n.hip)
    want = source if source.endswith(".o") else source.replace(".hip", ".o")
    target = [(o, c) for o, c in jobs if os.path.basename(o) == want]
    assert len(target) == 1, "no single object for " + source
    obj, cmd = target[0]
    src_path = cmd[cmd.index("-c") + 1]
    source = os.path.basename(src_path)
    procs = []
    for spec in specs:
        name, rest = spec.split("=", 1)
        parts = rest.split(",-D")
        variant, flags = parts[0], ["-D" + f for f in parts[1:]]
        vsrc = os.path.join(_build.CSRC, "_ab_%s_%s" % (name, source))
        shutil.copyfile(variant, vsrc)
        o2 = obj.replace(".o", ".ab_%s.o" % name)
        c2 = [o2 if c == obj else (vsrc if c == src_path else c) for c in cmd] + flags
        procs.append((name, o2, vsrc, subprocess.Popen(c2)))
    for name, o2, vsrc, pr in procs:
        assert pr.wait() == 0, name
        os.remove(vsrc)
        so = os.path.join(_build.CSRC, "libdqp_hip_ab_%s.so" % name)
        objs = [o2 if o == obj else o for o, _ in jobs]
        subprocess.check_call([_build.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
        print("built", so)


if __name__ == "__main__":
    assert len(sys.argv) > 3 and sys.argv[1] == "build", __doc__
    build(sys.argv[2], sys.argv[3:])
