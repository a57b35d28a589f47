#!/usr/bin/env python3
"""Regenerates tests/golden/* from the COMPILED reference (oracle/_ref, fixed-point tree).
Runs only where /root/reference was available to build oracle/_ref (`make -C oracle ref`).

Fixtures (data only -- inputs and expected outputs):
  Ch_f1_raw.pcm        the reference's own 16 kHz test input (JC1_SDK_SRC_ARM/bin/Ch_f1_raw.pcm)
  ch_f1.bit            reference encoder output for it, in the reference CLI's .bit container
  synth8x25.npz        8 synthetic streams x 25 packets: pcm, reference bitstreams, reference decodes
                       (clean and with a fixed per-description loss mask)
  golden.json          md5 known answers (bitstream, decoded PCM at 0 % and 30 % CLI loss, ...)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refcodec as R  # noqa: E402
import solo_testlib as T  # noqa: E402


def md5(b):
    return hashlib.md5(bytes(b)).hexdigest()


def cold_recv_mask(recv):
    r = recv.copy()
    for i in range(r.shape[0]):
        k = 1 + (i * 5) % 6
        r[i, :k] = 0
        r[i, k] = (3, 1, 2)[i % 3]
    return r


def main():
    g = {}
    pcm = np.fromfile(os.path.join(HERE, "Ch_f1_raw.pcm"), np.int16)
    g["ch_f1_pcm_md5"] = md5(pcm.tobytes())
    npk = pcm.size // 640
    enc = R.RefEncoder("fix")
    recs = [enc.encode(pcm[p * 640:(p + 1) * 640]) for p in range(npk)]
    bit = T.write_bit_container(recs)
    open(os.path.join(HERE, "ch_f1.bit"), "wb").write(bit)
    g["ch_f1_bit_md5"] = md5(bit)
    g["ch_f1_packets"] = npk
    for loss in (0, 30):
        dec = R.RefDecoder("fix")
        pat = R.cli_loss_pattern(npk, loss)
        out = bytearray()
        for p, (pl, n0, n1) in enumerate(recs):
            x, ret = dec.decode(*R.map_loss(pl, n0, n1, *pat[p]))
            out += x.tobytes()
        g["ch_f1_dec_loss%d_md5" % loss] = md5(out)
    # FLP tree on the same input (PCM-tolerance anchor, not bit-exact target)
    encf = R.RefEncoder("flp")
    recsf = [encf.encode(pcm[p * 640:(p + 1) * 640]) for p in range(npk)]
    g["ch_f1_flp_bit_bytes"] = len(T.write_bit_container(recsf))

    # synthetic streams
    N, P = 8, 25
    spcm = np.stack([R.synth_stream(i, P) for i in range(N)])
    streams = []
    for i in range(N):
        e = R.RefEncoder("fix")
        streams.append([e.encode(spcm[i, p]) for p in range(P)])
    bits, nb = T.pack_slots(streams)
    recv = T.bernoulli_recv(N, P, 0.3, 1234)
    dec_clean = np.zeros((N, P, 640), np.int16)
    dec_loss = np.zeros((N, P, 640), np.int16)
    for i in range(N):
        d0, d1 = R.RefDecoder("fix"), R.RefDecoder("fix")
        for p, (pl, n0, n1) in enumerate(streams[i]):
            dec_clean[i, p], _ = d0.decode(pl, n0, n1, 4)
            m = int(recv[i, p])
            dec_loss[i, p], _ = d1.decode(*R.map_loss(pl, n0, n1, not (m & 1), not (m & 2)))
    np.savez_compressed(os.path.join(HERE, "synth8x25.npz"), pcm=spcm, bits=bits, nbytes=nb, recv=recv,
                        dec_clean=dec_clean, dec_loss=dec_loss)
    # decoder cold start: the first 1..6 packets of each stream never arrive (the reference decoder is still at its initial
    # 24 kHz then), the first packet that does arrive is complete / MD1 only / MD2 only by stream, later ones per `recv`
    recv_cold = cold_recv_mask(recv)
    dec_cold = np.zeros((N, P, 640), np.int16)
    for i in range(N):
        d = R.RefDecoder("fix")
        for p, (pl, n0, n1) in enumerate(streams[i]):
            m = int(recv_cold[i, p])
            dec_cold[i, p], ret = d.decode(*R.map_loss(pl, n0, n1, not (m & 1), not (m & 2)))
            assert ret == 0
    np.savez_compressed(os.path.join(HERE, "synth8x25_cold.npz"), recv=recv_cold, dec=dec_cold)
    g["synth_dec_cold_md5"] = md5(dec_cold.tobytes())
    # 32 kHz mode (decoder): 4 streams x 20 packets of 1280 samples at 24 kbps through the reference encoder (SILK wide band),
    # reference decodes clean and with description loss (stream 1 starts with two lost packets, stream 3 is joint_mode 1)
    NW, PW = 4, 20
    wpcm = np.stack([T.synth_stream_32k(500 + i, PW) for i in range(NW)])
    wstreams = []
    for i in range(NW):
        e = R.RefEncoder("fix", rate=24000, samplerate=32000, joint=1 if i == 3 else 0)
        wstreams.append([e.encode(wpcm[i, p]) for p in range(PW)])
    wbits, wnb = T.pack_slots(wstreams)
    wrecv = T.bernoulli_recv(NW, PW, 0.3, 4321)
    wrecv[1, :2] = 0
    wdec_clean = np.zeros((NW, PW, 1280), np.int16)
    wdec_loss = np.zeros((NW, PW, 1280), np.int16)
    for i in range(NW):
        d0, d1 = (R.RefDecoder("fix", samplerate=32000, joint=1 if i == 3 else 0) for _ in range(2))
        for p, (pl, n0, n1) in enumerate(wstreams[i]):
            wdec_clean[i, p], r0 = d0.decode(pl, n0, n1, 4)
            m = int(wrecv[i, p])
            wdec_loss[i, p], r1 = d1.decode(*R.map_loss(pl, n0, n1, not (m & 1), not (m & 2)))
            assert r0 == 0 and r1 == 0
    np.savez_compressed(os.path.join(HERE, "wb4x20.npz"), pcm=wpcm, bits=wbits, nbytes=wnb, recv=wrecv, dec_clean=wdec_clean, dec_loss=wdec_loss)
    g["wb_bits_md5"] = md5(wbits.tobytes())
    g["wb_dec_clean_md5"] = md5(wdec_clean.tobytes())
    g["wb_dec_loss_md5"] = md5(wdec_loss.tobytes())
    g["wb_mean_payload"] = float(wnb[..., 0].mean())
    # file-level known answers of the 32 kHz mode (`-Fs_API 32000 -rate 24000`): .bit container of stream 0 of the batch above and
    # its decode at 0 % and 30 % CLI loss
    kat_bit = T.write_bit_container(wstreams[0])
    g["wb_kat_bit_md5"] = md5(kat_bit)
    for loss in (0, 30):
        dec = R.RefDecoder("fix", samplerate=32000)
        pat = R.cli_loss_pattern(PW, loss, [(r[1], r[2]) for r in wstreams[0]])
        outp = bytearray()
        for p, (pl, n0, n1) in enumerate(wstreams[0]):
            x, ret = dec.decode(*R.map_loss(pl, n0, n1, *pat[p]))
            outp += x.tobytes()
        g["wb_kat_dec_loss%d_md5" % loss] = md5(outp)
    g["synth_bits_md5"] = md5(bits.tobytes())
    g["synth_dec_clean_md5"] = md5(dec_clean.tobytes())
    g["synth_dec_loss_md5"] = md5(dec_loss.tobytes())
    g["synth_mean_payload"] = float(nb[..., 0].mean())
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(g, f, indent=1, sort_keys=True)
    print(json.dumps(g, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
