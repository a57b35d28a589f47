"""The row exchanges of the quantiser kernel (solo_amd/csrc/solo_enc_nsq_row.h) on the GPU against their host definitions:
bank-masked DPP row shifts (track -> track), row rotations (sums / ors over the tracks), quad permutes (state -> state)."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _host(v, idx):
    v = v.astype(np.int64)
    out = np.zeros((14, 64), np.int64)
    for lane in range(64):
        row, l = lane & ~15, lane & 15
        t, k = l >> 2, l & 3
        quad = lane & ~3
        for T in range(3):
            out[T, lane] = v[row + 4 * T + k]
        out[3, lane] = v[row + 4 + k] if t == 0 else -1
        out[4, lane] = v[row + 8 + k] if t == 0 else -1
        tr = [v[row + 4 * q + k] for q in range(3)]
        out[5, lane] = sum(tr)
        out[6, lane] = int(tr[0]) | int(tr[1]) | int(tr[2])
        out[7, lane] = v[quad + idx[lane]]
        q4 = v[quad:quad + 4]
        out[8, lane], out[9, lane] = q4.min(), int(np.argmin(q4))
        out[10, lane], out[11, lane] = q4.max(), int(np.argmax(q4))
        out[12, lane] = q4.sum()
        out[13, lane] = v[quad + idx[lane]]
    return ((out + 2**31) % 2**32 - 2**31).astype(np.int32)


def test_row_exchanges():
    import torch
    import solo_amd
    lib = solo_amd.load_library()
    lib.solo_debug_rowops.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.solo_debug_rowops.restype = C.c_int
    rng = np.random.default_rng(5)
    for trial in range(8):
        if trial < 4:
            v = rng.integers(-2**31, 2**31, 64, dtype=np.int64).astype(np.int32)
        else:                                # ties: the first index must win
            v = rng.integers(0, 3, 64).astype(np.int32)
        idx = rng.integers(0, 4, 64).astype(np.int32)
        d_in, d_idx = torch.from_numpy(v).cuda(), torch.from_numpy(idx).cuda()
        d_out = torch.zeros(14 * 64, dtype=torch.int32, device="cuda")
        assert lib.solo_debug_rowops(d_in.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), None) == 0
        torch.cuda.synchronize()
        got = d_out.cpu().numpy().reshape(14, 64)
        want = _host(v, idx)
        lanes = np.arange(64)
        live = ((lanes >> 2) & 3) < 3                   # the spare quad of every row receives nothing defined from the track exchanges
        centre = ((lanes >> 2) & 3) == 0
        for r in range(14):
            m = live if r < 3 else (centre if r in (3, 4) else np.ones(64, bool))
            assert np.array_equal(got[r][m], want[r][m]), (trial, r, got[r].tolist(), want[r].tolist())
