#!/usr/bin/env python3
"""Regenerates tests/golden/nb_stale_coder.npz from the COMPILED reference (oracle/_ref): trial 299538 of
tools/debug/fuzz_decoder_gen.py (16 kHz mode, 24 kbps, 14 packets).  A two-description packet decoded through the batch path's
records (packet 6) is followed, several packets later, by a corrupted single-description packet that announces more frames than it
carries (packet 11), a lost packet, and a two-description packet (13) whose first call therefore decodes on in the OLD buffers of
BOTH description slots -- the second slot's coder registers date from packet 6 (sx_decode_packet rebuilds them from the shadow
buffer).  The reference rejects packet 13 (-12).  Data only: the packets as received (corrupted bytes included), receive masks,
reference PCM and return codes; runs where /root/reference was available to build oracle/_ref."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools", "debug")):
    sys.path.insert(0, p)
import refcodec as R  # noqa: E402
import fuzz_decoder_gen as F  # noqa: E402

SEED = 299538


def main():
    cfg, seq = F.sequence(SEED)
    assert not cfg["wb"] and cfg["mdi"] == 0 and cfg["joint"] == 0, cfg
    P = len(seq)
    slot = 256
    bits = np.zeros((1, P, slot), np.uint8)
    nb = np.zeros((1, P, 2), np.int16)
    recv = np.zeros((1, P), np.uint8)
    dec = np.zeros((1, P, 640), np.int16)
    ret = np.zeros((1, P), np.int32)
    d = R.RefDecoder("fix")
    for p, (a, hit, (pl, n0, n1, m)) in enumerate(seq):
        bits[0, p, :n0] = np.frombuffer(pl, np.uint8)
        nb[0, p] = (n0, n1)
        recv[0, p] = m
        assert a == (R.map_loss(pl, n0, n1, not (m & 1), not (m & 2)) if m else (pl, n0, n1, 1))
        dec[0, p], ret[0, p] = d.decode(*a)
        if ret[0, p] < 0:
            bits, nb, recv, dec, ret = bits[:, :p + 1], nb[:, :p + 1], recv[:, :p + 1], dec[:, :p + 1], ret[:, :p + 1]
            break
    assert ret[0, -1] == -12 and (ret[0, :-1] == 0).all(), ret
    np.savez_compressed(os.path.join(HERE, "nb_stale_coder.npz"), bits=bits, nbytes=nb, recv=recv, dec=dec, ret=ret)
    print("nb_stale_coder.npz: %d packets, receive masks %s" % (bits.shape[1], recv[0].tolist()))


if __name__ == "__main__":
    main()
