#!/usr/bin/env python3
"""Regenerates tests/golden/wb_more_frames.npz from the COMPILED reference (oracle/_ref): the one known input on which the
reference decoder's output high-pass runs (SKP_Silk_decode_frame.c:381: nFramesDecoded > 2).  Stream 0 of wb4x20.npz (32 kHz mode),
byte 56 of packet 4 changed from 74 to 212, descriptions received as MD1 / MD2 / both / both / MD2 / MD1 / ...: the range decoder
accepts the corrupted second description of packet 4, whose termination symbol announces more frames than the packet carries, so
packet 5's first call decodes on in packet 4's buffer as "frame 3" and its output is high-pass filtered (found by
tools/debug/fuzz_decoder_emu.py ... wb).  Data only; runs where /root/reference was available to build oracle/_ref."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import refcodec as R  # noqa: E402

RECV = [1, 2, 3, 3, 2, 1, 3]           # bit 0: MD1 arrived, bit 1: MD2 (+ high band) arrived.  Packet 6 is rejected by the reference (-12: its
                                       # calls are still out of step with the packets); its state after a rejection is not defined by its inputs


def main():
    z = np.load(os.path.join(HERE, "wb4x20.npz"))
    P = len(RECV)
    bits = z["bits"][:1, :P].copy()
    nb = z["nbytes"][:1, :P].copy()
    assert bits[0, 4, 56] == 74
    bits[0, 4, 56] = 212
    recv = np.array([RECV], np.uint8)
    d = R.RefDecoder("fix", samplerate=32000)
    dec = np.zeros((1, P, 1280), np.int16)
    rets = np.zeros((1, P), np.int32)
    for p in range(P):
        n0, n1, m = int(nb[0, p, 0]), int(nb[0, p, 1]), int(recv[0, p])
        dec[0, p], rets[0, p] = d.decode(*R.map_loss(bits[0, p, :n0].tobytes(), n0, n1, not (m & 1), not (m & 2)))
    assert (rets[0, :P - 1] == 0).all() and rets[0, P - 1] < 0, rets
    np.savez_compressed(os.path.join(HERE, "wb_more_frames.npz"), bits=bits, nbytes=nb, recv=recv, dec=dec, ret=rets)
    print("wb_more_frames.npz written")


if __name__ == "__main__":
    main()
