#!/usr/bin/env python3
"""Regenerates tests/golden/edge28x12.npz (and edge_wb13x8.npz, the same for the 32 kHz mode) from the COMPILED reference (oracle/_ref, fixed-point tree): 28 un-speech-like streams
(two of every family of solo_amd.synth.edge_stream: silence with stray LSBs, full-scale noise / square waves / sweeps, DC,
impulses, 90 dB level ramps, clipped and very quiet signals, bursts, high-band-only tones, the Nyquist pattern, sub-audio sines,
random walks) x 12 packets: input PCM, reference bitstreams, CRC-32 of every reference-decoded packet under a fixed per-description
loss mask.  Data only; runs where /root/reference was available to build oracle/_ref (`make -C oracle ref`)."""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import refcodec as R  # noqa: E402
import solo_testlib as T  # noqa: E402
from solo_amd.synth import EDGE_FAMILIES, edge_stream  # noqa: E402


def main():
    N, P = 2 * EDGE_FAMILIES, 12
    pcm = np.stack([edge_stream(i, P) for i in range(N)])
    streams = []
    for i in range(N):
        e = R.RefEncoder("fix")
        streams.append([e.encode(pcm[i, p]) for p in range(P)])
    bits, nb = T.pack_slots(streams)
    bits = np.ascontiguousarray(bits[:, :, :int(nb[..., 0].max())])
    recv = T.bernoulli_recv(N, P, 0.25, 777)
    recv[:, 0] = 3
    crc = np.zeros((N, P), np.uint32)
    for i in range(N):
        d = R.RefDecoder("fix")
        for p, (pl, n0, n1) in enumerate(streams[i]):
            m = int(recv[i, p])
            x, ret = d.decode(pl, n0, n1, 1) if m == 0 else d.decode(*R.map_loss(pl, n0, n1, not (m & 1), not (m & 2)))
            assert ret == 0
            crc[i, p] = zlib.crc32(x.tobytes())
    np.savez_compressed(os.path.join(HERE, "edge28x12.npz"), pcm=pcm, bits=bits, nbytes=nb, recv=recv, dec_crc=crc)
    print("edge28x12.npz: mean payload %.1f B, max %d B" % (float(nb[..., 0].mean()), int(nb[..., 0].max())))
    # 32 kHz mode (1280-sample packets, 24 kbps): one stream per family except family 2 -- full-scale square waves make the compiled
    # reference overflow its stack at 32 kHz, so there is no reference answer for them
    fams = [f for f in range(EDGE_FAMILIES) if f != 2]
    PW = 8
    wpcm = np.stack([edge_stream(EDGE_FAMILIES * 3 + f, 2 * PW).reshape(PW, 1280) for f in fams])
    wstreams = []
    for i in range(len(fams)):
        e = R.RefEncoder("fix", rate=24000, samplerate=32000)
        wstreams.append([e.encode(wpcm[i, p]) for p in range(PW)])
    wbits, wnb = T.pack_slots(wstreams, slot=1024)
    wbits = np.ascontiguousarray(wbits[:, :, :int(wnb[..., 0].max())])
    wrecv = T.bernoulli_recv(len(fams), PW, 0.25, 778)
    wcrc = np.zeros((len(fams), PW), np.uint32)
    for i in range(len(fams)):
        d = R.RefDecoder("fix", samplerate=32000)
        for p, (pl, n0, n1) in enumerate(wstreams[i]):
            m = int(wrecv[i, p])
            x, ret = d.decode(pl, n0, n1, 1) if m == 0 else d.decode(*R.map_loss(pl, n0, n1, not (m & 1), not (m & 2)))
            assert ret == 0
            wcrc[i, p] = zlib.crc32(x.tobytes())
    np.savez_compressed(os.path.join(HERE, "edge_wb13x8.npz"), pcm=wpcm, bits=wbits, nbytes=wnb, recv=wrecv, dec_crc=wcrc)
    print("edge_wb13x8.npz: mean payload %.1f B, max %d B" % (float(wnb[..., 0].mean()), int(wnb[..., 0].max())))


if __name__ == "__main__":
    main()
