#!/usr/bin/env python3
"""Register / scratch / LDS metadata of every kernel in the built objects (csrc/*.o): unbundles
the gfx950 code object and reads the amdhsa notes.  usage: kernel_resources.py [object ...]"""
import glob, os, re, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
objs = sys.argv[1:] or sorted(glob.glob(os.path.join(HERE, "diff-qp-mpc_amd", "csrc", "*.o")))
for o in objs:
    with tempfile.TemporaryDirectory() as td:
        co, fat = os.path.join(td, "k.co"), os.path.join(td, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, o],
                       capture_output=True)
        if not os.path.exists(fat):
            continue
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co],
                           capture_output=True, text=True)
        if r.returncode or not os.path.exists(co) or os.path.getsize(co) == 0:
            continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    for blk in notes.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
        agpr = blk.split()[0]
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        print("%-28s vgpr %3s agpr %3s spill %4s scratch %6s B lds %6s  %s" % (
            os.path.basename(o), g("vgpr_count"), agpr, g("vgpr_spill_count"), g("private_segment_fixed_size"),
            g("group_segment_fixed_size"), name[:90]))
