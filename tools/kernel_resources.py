#!/usr/bin/env python3
"""Registers / scratch / LDS of the kernels inside a built library (default: the in-tree one):  python tools/kernel_resources.py [lib.so] [name fragment]"""
import os, re, shutil, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "solo_amd", "libsolo_mi355x.so")
frag = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as d:
    shutil.copy(lib, os.path.join(d, "lib.so"))
    subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=d)
    for f in sorted(os.listdir(d)):
        if "amdgcn" not in f:
            continue
        notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(d, f)], text=True)
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            if frag not in name:
                continue
            g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
            print("%-44s vgpr %3d + agpr %3d  sgpr %3d  scratch %4d B/lane  LDS %6d B" % (name, g("vgpr_count"), int(blk.split()[0]), g("sgpr_count"),
                                                                                       g("private_segment_fixed_size"), g("group_segment_fixed_size")))
