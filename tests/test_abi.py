"""The C-ABI library builds, loads, and exports every symbol include/solo_mi355x.h declares.
No compute call is made here (there is no GPU in the CPU test environment)."""
import ctypes as C
import os
import re

import pytest

import solo_amd
import solo_testlib as T


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(solo_amd.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return C.CDLL(solo_amd.LIB_PATH)


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(T.ROOT, "include", "solo_mi355x.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(AGR_Sate_\w+|solo_\w+)\s*\(", hdr))
    assert names == set(solo_amd.ABI_SYMBOLS)
    for n in sorted(names):
        assert hasattr(lib, n), n


def test_struct_layouts_match_reference_header():
    # AGR_JC1_SDK_API.h:11-31 -- 8 and 6 consecutive int32
    assert C.sizeof(solo_amd.USER_Ctrl_enc) == 32
    assert C.sizeof(solo_amd.USER_Ctrl_dec) == 24


def test_product_does_not_touch_oracle():
    """The product tree must not import / include / link anything under oracle/ or tests/."""
    bad = []
    for root, _, files in os.walk(os.path.join(T.ROOT, "solo_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".inc", ".cpp")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                if re.search(r"(#include|import|from)\s+[\"<]?[\w./]*(oracle|refcodec|tests/)", txt):
                    bad.append(f)
    assert not bad, bad


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        solo_amd.SoloBatch(4)
