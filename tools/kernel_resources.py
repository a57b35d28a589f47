#!/usr/bin/env python3
"""Registers / scratch / LDS of the kernels inside a built library (default: the in-tree one):  python tools/kernel_resources.py [lib.so] [name fragment]
Two sources per kernel: the metadata notes (what the compiler counted) and the kernel descriptor (what the hardware ALLOCATES per wavefront:
(granulated count + 1) x 8 -- the backend pads it up when it believes the occupancy cannot be higher anyway; the descriptor is what decides
who fits beside whom on a SIMD)."""
import os, re, shutil, struct, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_resources(lib):
    """{kernel name: dict(vgpr, agpr, sgpr, scratch, lds, vgpr_alloc)}"""
    res = {}
    with tempfile.TemporaryDirectory() as d:
        shutil.copy(lib, os.path.join(d, "lib.so"))
        subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=d)
        for f in sorted(os.listdir(d)):
            if "amdgcn" not in f:
                continue
            p = os.path.join(d, f)
            notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", p], text=True)
            for blk in notes.split("- .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
                res[name] = dict(vgpr=g("vgpr_count"), agpr=int(blk.split()[0]), sgpr=g("sgpr_count"), scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"), vgpr_alloc=None)
            sect = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "-S", "-W", p], text=True)
            ro = None
            for line in sect.splitlines():
                q = line.split()
                if ".rodata" in q:
                    i = q.index(".rodata"); ro = (int(q[i + 2], 16), int(q[i + 3], 16))
            if ro is None:
                continue
            blob = open(p, "rb").read()
            for line in subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "-s", "-W", p], text=True).splitlines():
                q = line.split()
                if len(q) >= 8 and q[-1].endswith(".kd") and q[-1][:-3] in res:
                    off = ro[1] + int(q[1], 16) - ro[0]
                    rsrc1 = struct.unpack_from("<I", blob, off + 48)[0]
                    res[q[-1][:-3]]["vgpr_alloc"] = ((rsrc1 & 0x3f) + 1) * 8
    return res


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "solo_amd", "libsolo_mi355x.so")
    frag = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, r in kernel_resources(lib).items():
        if frag in name:
            print("%-44s vgpr %3d + agpr %3d (allocated per wavefront: %3s)  sgpr %3d  scratch %4d B/lane  LDS %6d B" % (
                name[:44], r["vgpr"], r["agpr"], r["vgpr_alloc"], r["sgpr"], r["scratch"], r["lds"]))
