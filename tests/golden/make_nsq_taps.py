#!/usr/bin/env python3
"""Makes tests/golden/nsq_taps.npz: the ARGUMENTS and OUTPUTS of every SKP_Silk_NSQ_del_dec call (SKP_Silk_NSQ_del_dec.c:925) the compiled
reference makes while it encodes (a) the first 100 packets of its own speech sample Ch_f1_raw.pcm and (b) 100 packets of a synthetic
stream -- 2 x 200 quantiser calls.  Needs oracle/_ref/libsolo_ref_fix_taps.so (`make -C oracle taps`: the unmodified reference linked
with oracle/ref_taps.c through -Wl,--wrap), i.e. runs in the build container only; the file it writes is data (inputs and expected
outputs) and travels.  tests/test_nsq_taps.py feeds the recorded arguments to the quantiser kernel ALONE and compares."""
import ctypes as C
import os
import sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT]
import refcodec as R
import solo_testlib as T

lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libsolo_ref_fix_taps.so"))
SZ_IN, SZ_OUT = lib.solo_nsq_tap_sizeof_in(), lib.solo_nsq_tap_sizeof_out()
assert SZ_IN == 660 and SZ_OUT == 4 + 320 + 640, (SZ_IN, SZ_OUT)
tap_n = C.c_int.in_dll(lib, "solo_nsq_tap_n")
tap_in = (C.c_ubyte * (512 * SZ_IN)).in_dll(lib, "solo_nsq_tap_in")
tap_out = (C.c_ubyte * (512 * SZ_OUT)).in_dll(lib, "solo_nsq_tap_out")


def run(pcm):                     # [P, 640] int16 -> (in [2P, 660] u8, out [2P, 964] u8)
    tap_n.value = 0
    ctrl = R.default_enc_ctrl()
    lib.AGR_Sate_Encoder_Init.restype = C.c_void_p
    lib.AGR_Sate_Encoder_Init.argtypes = [C.c_void_p]
    lib.AGR_Sate_Encoder_Encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    h = lib.AGR_Sate_Encoder_Init(C.byref(ctrl))
    bits, nb = np.zeros(2048, np.uint8), np.zeros(6, np.int16)
    for p in range(pcm.shape[0]):
        x = np.ascontiguousarray(pcm[p])
        lib.AGR_Sate_Encoder_Encode(h, x.ctypes.data, bits.ctypes.data, 2048, nb.ctypes.data)
    n = tap_n.value
    assert n == 2 * pcm.shape[0], n
    return (np.frombuffer(tap_in, np.uint8, n * SZ_IN).reshape(n, SZ_IN).copy(), np.frombuffer(tap_out, np.uint8, n * SZ_OUT).reshape(n, SZ_OUT).copy())


P = 100
speech = T.load_ch_f1()[:P * 640].reshape(P, 640)
synth = R.synth_stream(4711, P)
ins, outs = zip(*(run(x) for x in (speech, synth)))
np.savez_compressed(os.path.join(HERE, "nsq_taps.npz"), nsq_in=np.stack(ins), nsq_out=np.stack(outs),
                    note=np.array("streams: Ch_f1_raw.pcm packets 0..99, synth_stream(4711); per stream 200 calls; nsq_in rows = struct SxNsqIn (660 B), "
                                  "nsq_out rows = {int32 Seed; int8 q[2][160]; int32 r[160]} of the reference"))
print("wrote nsq_taps.npz:", np.stack(ins).shape, np.stack(outs).shape)
