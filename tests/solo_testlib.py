"""Shared helpers of the test-suite (data plumbing only -- no codec arithmetic lives here)."""
import ctypes as C
import hashlib
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
SLOT = 512


def md5(a):
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()


def golden_json():
    with open(os.path.join(GOLDEN, "golden.json")) as f:
        return json.load(f)


def load_ch_f1():
    return np.fromfile(os.path.join(GOLDEN, "Ch_f1_raw.pcm"), np.int16)


def parse_bit_container(raw):
    """The reference CLI's .bit container (test/enc_main.c:243-249): per packet
    int16 total, int16 len(MD2)+8, then `total` payload bytes."""
    recs, pos = [], 0
    raw = bytes(raw)
    while pos + 4 <= len(raw):
        n0, n1 = np.frombuffer(raw[pos:pos + 4], np.int16)
        pos += 4
        recs.append((raw[pos:pos + int(n0)], int(n0), int(n1)))
        pos += int(n0)
    return recs


def write_bit_container(recs):
    out = bytearray()
    for pl, n0, n1 in recs:
        out += np.array([n0, n1], np.int16).tobytes() + bytes(pl[:n0])
    return bytes(out)


def pack_slots(streams, slot=SLOT):
    """streams: list (per stream) of lists of (payload, n0, n1) -> bits [N,P,slot] uint8, nbytes [N,P,2] int16"""
    N, P = len(streams), len(streams[0])
    bits = np.zeros((N, P, slot), np.uint8)
    nb = np.zeros((N, P, 2), np.int16)
    for s, recs in enumerate(streams):
        for p, (pl, n0, n1) in enumerate(recs):
            bits[s, p, :n0] = np.frombuffer(pl[:n0], np.uint8)
            nb[s, p] = (n0, n1)
    return bits, nb


def recv_mask_from_pattern(pattern):
    """[(lost_md1, lost_md2)] -> uint8 mask, bit0 = MD1 arrived, bit1 = MD2 arrived"""
    return np.array([(0 if l1 else 1) | (0 if l2 else 2) for l1, l2 in pattern], np.uint8)


def bernoulli_recv(n_streams, n_packets, p_loss, seed):
    """Config-4 style mask: per (stream, packet, description) Bernoulli(p_loss) drop, first packet received."""
    rng = np.random.default_rng(seed)
    lost = rng.random((n_streams, n_packets, 2)) < p_loss
    lost[:, 0, :] = False
    return ((~lost[..., 0]).astype(np.uint8) | ((~lost[..., 1]).astype(np.uint8) << 1)).astype(np.uint8)


def synth_stream_32k(seed, n_packets):
    """32 kHz test material [n_packets, 1280]: the 16 kHz workload generator read at twice the rate (pitch and formants double,
    which is fine for exercising the codec: voiced / unvoiced / silence all occur)."""
    from solo_amd.synth import synth_stream
    return synth_stream(seed, 2 * n_packets).reshape(n_packets, 1280)


# ---- host emulation of the kernel source (tests/emu) ------------------------------------------------
_emu = None
_emu_wb = None


def load_emu():
    global _emu
    if _emu is None:
        d = os.path.join(ROOT, "tests", "emu")
        subprocess.check_call(["make", "-s", "-C", d])
        lib = C.CDLL(os.path.join(d, "libsolo_emu.so"))
        lib.emu_dec_create.restype = C.c_void_p
        lib.emu_dec_create.argtypes = [C.c_int]
        lib.emu_dec_destroy.argtypes = [C.c_void_p]
        lib.emu_dec_packet.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        if hasattr(lib, "emu_enc_create"):
            lib.emu_enc_create.restype = C.c_void_p
            lib.emu_enc_create.argtypes = [C.c_int, C.c_int]
            lib.emu_enc_destroy.argtypes = [C.c_void_p]
            lib.emu_enc_packet.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _emu = lib
    return _emu


def load_emu_wb():
    """32 kHz-mode build of the decoder source (-DSX_FS_KHZ=16): same entry points, 1280-sample packets."""
    global _emu_wb
    if _emu_wb is None:
        d = os.path.join(ROOT, "tests", "emu")
        subprocess.check_call(["make", "-s", "-C", d, "libsolo_emu_wb.so"])
        lib = C.CDLL(os.path.join(d, "libsolo_emu_wb.so"))
        lib.emu_dec_create.restype = C.c_void_p
        lib.emu_dec_create.argtypes = [C.c_int]
        lib.emu_dec_destroy.argtypes = [C.c_void_p]
        lib.emu_dec_packet.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        assert lib.emu_packet_samples() == 1280
        if hasattr(lib, "emu_enc_create"):
            lib.emu_enc_create.restype = C.c_void_p
            lib.emu_enc_create.argtypes = [C.c_int, C.c_int]
            lib.emu_enc_destroy.argtypes = [C.c_void_p]
            lib.emu_enc_packet.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _emu_wb = lib
    return _emu_wb


class EmuDecoder:
    # 1: decode like the batch path's two kernels (history-free symbol extraction per description -> records -> decoder);
    # 0: the decoder reads its symbols itself.  tests/test_emu_decoder.py runs its cases in both.
    SPLIT = int(os.environ.get("SOLO_EMU_DEC_SPLIT", "0"))

    def __init__(self, use_md_index=0, wb=False, split=None):
        self.lib = load_emu_wb() if wb else load_emu()
        self.n = 1280 if wb else 640
        self.h = self.lib.emu_dec_create(use_md_index)
        split = EmuDecoder.SPLIT if split is None else split
        if split:
            self.lib.emu_dec_set_split.argtypes = [C.c_void_p, C.c_int]
            self.lib.emu_dec_set_split(self.h, split)

    def decode(self, payload, n0, n1, lostflag):
        buf = np.zeros(1100, np.uint8)
        b = np.frombuffer(payload, np.uint8)
        buf[:b.size] = b
        out = np.zeros(self.n, np.int16)
        ret = self.lib.emu_dec_packet(self.h, buf.ctypes.data, n0, n1, lostflag, out.ctypes.data)
        return out, ret

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.emu_dec_destroy(self.h)
            self.h = None


class EmuEncoder:
    def __init__(self, rate=13600, use_md_index=0, wb=False):
        self.lib = load_emu_wb() if wb else load_emu()
        self.h = self.lib.emu_enc_create(rate, use_md_index)

    def encode(self, pcm640):
        pcm = np.ascontiguousarray(pcm640, np.int16)
        bits = np.zeros(1100, np.uint8)
        nb = np.zeros(2, np.int16)
        n = self.lib.emu_enc_packet(self.h, pcm.ctypes.data, bits.ctypes.data, 1024, nb.ctypes.data)
        return bits[:max(n, 0)].tobytes(), int(nb[0]), int(nb[1])

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.emu_enc_destroy(self.h)
            self.h = None
