"""Register / LDS budgets of the shipped kernels, read from the code objects inside solo_amd/libsolo_mi355x.so (no GPU needed).
The residency plan of a compute unit hangs on them: an analysis wave at 96 registers (five per SIMD, four
beside a quantiser wave), a quantiser wave at 128, the decoder's kernels at 128 (four per SIMD: sixteen workgroups in one round) --
one register more and the allocation granularity of 8 takes a whole wave per SIMD away without any test noticing (it happened to
the decoder when the encoder kernels moved to their own translation unit)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

import solo_amd

LLVM = "/opt/rocm/lib/llvm/bin"
BUDGET = {                        # kernel name fragment -> (max vector registers, max LDS bytes)
    "solo_enc_analysis_kernel": (96, 8704), "solo_enc_coding_kernel": (96, 8704), "solo_nsq_kernel": (128, 6144),
    "solo_dec_synth_kernel": (128, 10240), "solo_decode_kernel": (128, 10240), "solo_decode_split_kernel": (128, 10240),
    "solo_decode_ring_kernel": (128, 10240), "solo_dec_extract_kernel": (96, 14336),
}


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-objdump")), reason="no LLVM binutils on this box")
def test_vector_register_and_lds_budgets():
    assert os.path.exists(solo_amd.LIB_PATH)
    seen = {}
    with tempfile.TemporaryDirectory() as d:
        lib = os.path.join(d, "lib.so")
        shutil.copy(solo_amd.LIB_PATH, lib)
        subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", lib], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=d)
        for f in sorted(os.listdir(d)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(d, f)], text=True)
            for blk in notes.split("- .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                v = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1)) + int(blk.split()[0])
                lds = int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", blk).group(1))
                seen[name] = (v, lds)
    checked = 0
    for name, (v, lds) in seen.items():
        if "_wb" in name:                 # the 32 kHz build: larger frames, its own budgets are not part of the plan -- except the
            if "solo_dec_synth_kernel" in name or "solo_decode_kernel" in name:      # decoder's two waves per SIMD (round 5: 262 registers = one wave, decode 45 % slower)
                assert v <= 256, (name, "vector registers", v, "budget", 256)
            continue
        for frag, (vmax, lmax) in BUDGET.items():
            if frag in name:
                assert v <= vmax, (name, "vector registers", v, "budget", vmax)
                assert lds <= lmax, (name, "LDS bytes", lds, "budget", lmax)
                checked += 1
    assert checked >= len(BUDGET), (checked, sorted(seen))
