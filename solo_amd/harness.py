"""Host-side mirror of the reference's command-line harness: the `.bit` container written by
test/enc_main.c and read by test/dec_main.c, and the loss simulator of the decoder CLI, so that files produced here
interoperate with the stock reference binaries and their known-answer md5s can be matched end to end.

  python -m solo_amd.harness enc in.pcm out.bit [-rate bps] [-MDI 0/1] [-joint 1] [-DTX 1] [-Fs_API 32000] [-framesize 20]
  python -m solo_amd.harness dec in.bit out.pcm [-loss perc] [-dec_mode 1|2] [-MDI 0/1] [-joint 1] [-Fs_API 32000] [-framesize 20]

Record format (JC1_SDK_SRC_ARM/test/enc_main.c:243-249): per packet (40 ms; 20 ms with -framesize 20)  int16 total,
int16 len(MD2)+HB, `total` payload bytes (payload = MD1 || MD2 || HB; HB = 8 bytes, 4 with -framesize 20 or -joint 1).  Loss simulator (test/dec_main.c:24,236-252): rand_seed = 1, the LCG
seed <- 907633515 + seed * 196314165 (mod 2^32) is drawn twice (MD1, MD2) on every EVEN packet and the pair of decisions is
reused for the following odd packet; a description is lost when ((seed >> 16) + 32768) / 65535 < loss / 100 in float32.
The receiver-side mapping of the decisions to (ptr, nBytes, lostflag) (dec_main.c:255-378) runs on the GPU inside
solo_decode_kernel; this module only builds the per-packet mask (bit0 = MD1 arrived, bit1 = MD2 arrived).

Files are processed as ONE stream of P packets in one batched call (the GPU is meant for thousands of streams; this is the
conformance path)."""
import sys

import numpy as np

PACKET_SAMPLES = 640


def parse_bit_container(raw):
    """bytes -> [(payload, total, len(MD2)+8)] per packet"""
    recs, pos = [], 0
    raw = bytes(raw)
    while pos + 4 <= len(raw):
        n0, n1 = (int(v) for v in np.frombuffer(raw[pos:pos + 4], np.int16))
        pos += 4
        if n0 < 0 or pos + n0 > len(raw):
            raise ValueError("truncated .bit container at byte %d" % (pos - 4))
        recs.append((raw[pos:pos + n0], n0, n1))
        pos += n0
    return recs


def write_bit_container(recs):
    out = bytearray()
    for pl, n0, n1 in recs:
        out += np.array([n0, n1], np.int16).tobytes() + bytes(pl[:n0])
    return bytes(out)


def cli_loss_pattern(n_packets, loss_perc, nbytes=None):
    """[(lost_md1, lost_md2)] per packet, exactly the draws of `dec_main -loss P`.  nbytes (optional): [(total, len(MD2)+HB)] per
    packet -- an empty (DTX) packet drawn on an even packet counts as lost, like the CLI's `counter > 0` / `nBytes[j] == 0` tests."""
    seed, lost, out = 1, [0, 0], []
    thr = np.float32(loss_perc) / np.float32(100.0)
    for p in range(n_packets):
        if p % 2 == 0:
            for j in range(2):
                seed = (907633515 + seed * 196314165) & 0xFFFFFFFF
                s = seed - (1 << 32) if seed & 0x80000000 else seed
                v = np.float32((s >> 16) + (1 << 15)) / np.float32(65535.0)
                empty = nbytes is not None and (nbytes[p][0] <= 0 or nbytes[p][j] == 0)
                lost[j] = 0 if (v >= thr and not empty) else 1
        out.append(tuple(lost))
    return out


def recv_mask(pattern):
    """[(lost_md1, lost_md2)] -> uint8 mask for solo_batch_decode: bit0 = MD1 arrived, bit1 = MD2 arrived"""
    return np.array([(0 if l1 else 1) | (0 if l2 else 2) for l1, l2 in pattern], np.uint8)


def encode_pcm(pcm, rate=13600, use_md_index=0, slot_bytes=1088, joint=0, dtx=0, samplerate=16000, framesize_ms=40):
    """int16 array (16 kHz mono, or 32 kHz with samplerate=32000) -> [(payload, total, len(MD2)+8)]; a trailing partial packet is dropped like the CLI does"""
    import torch
    from . import SoloBatch
    pcm = np.asarray(pcm, np.int16)
    L = PACKET_SAMPLES * samplerate // 16000 * framesize_ms // 40
    P = pcm.size // L
    if P == 0:
        return []
    b = SoloBatch(1, rate=rate, encoder=True, decoder=False, slot_bytes=slot_bytes, use_md_index=use_md_index, joint=joint, dtx=dtx,
                  samplerate=samplerate, framesize_ms=framesize_ms)
    x = torch.from_numpy(np.ascontiguousarray(pcm[:P * L].reshape(1, P, L))).to(b.device)
    bits, nb, st = b.encode(x)
    torch.cuda.synchronize()
    if int(st[0]) != 0:
        raise RuntimeError("encoder status %d" % int(st[0]))
    hb, hn = bits.cpu().numpy()[0], nb.cpu().numpy()[0]
    return [(hb[p, :hn[p, 0]].tobytes(), int(hn[p, 0]), int(hn[p, 1])) for p in range(P)]


def decode_records(recs, loss_perc=0, use_md_index=0, slot_bytes=1088, joint=0, samplerate=16000, dec_mode=0, framesize_ms=40):
    """[(payload, total, len(MD2)+8)] -> int16 PCM, with the CLI's loss simulation (samplerate 32000 = `-Fs_API 32000`).

    dec_mode (test/dec_main.c:123-145,341-361): 1 = decode ONLY the first description of every packet of the file, 2 = only the
    second one (single-description decoding for the whole file: lostflag 2 / 3 throughout); like the CLI, it cannot be combined
    with a simulated loss.

    Empty records (total == 0: DTX packets the encoder did not send) are handled like the reference CLI does: the library call
    returns -1 without touching the decoder (AGR_BWE_SDK_API.c:266) and the CLI writes its output buffer again unchanged
    (test/dec_main.c:365-381), i.e. the previous packet's PCM is repeated and the decoder state does not move.  (The batched
    device API on its own would conceal an empty record as a lost packet; see include/solo_mi355x.h.)  The file is therefore
    decoded in runs of consecutive non-empty packets; the state carries across the calls."""
    import torch
    from . import SoloBatch
    P = len(recs)
    if P == 0:
        return np.zeros(0, np.int16)
    ns = PACKET_SAMPLES * samplerate // 16000 * framesize_ms // 40
    if dec_mode not in (0, 1, 2):
        raise ValueError("dec_mode: 0, 1 or 2")
    if dec_mode and loss_perc > 0:
        raise ValueError("loss and dec_mode can't be set at the same time")        # (dec_main.c:139-145)
    mask = recv_mask(cli_loss_pattern(P, loss_perc, [(r[1], r[2]) for r in recs]))
    if dec_mode:
        mask[:] = dec_mode          # the override of dec_main.c:341-361 = "only that description arrived", whatever the packet's parity
    b = SoloBatch(1, encoder=False, decoder=True, slot_bytes=slot_bytes, use_md_index=use_md_index, joint=joint, samplerate=samplerate, framesize_ms=framesize_ms)
    out = np.zeros((P, ns), np.int16)
    p = 0
    while p < P:
        if recs[p][1] <= 0:
            out[p] = out[p - 1] if p > 0 else 0
            p += 1
            continue
        e = p
        while e < P and recs[e][1] > 0:
            e += 1
        n = e - p
        bits = np.zeros((1, n, slot_bytes), np.uint8)
        nb = np.zeros((1, n, 2), np.int16)
        for k, (pl, n0, n1) in enumerate(recs[p:e]):
            if n0 > slot_bytes:
                raise ValueError("packet %d: %d bytes exceed the slot" % (p + k, n0))
            bits[0, k, :n0] = np.frombuffer(pl[:n0], np.uint8)
            nb[0, k] = (n0, n1)
        pcm, st = b.decode(torch.from_numpy(bits).to(b.device), torch.from_numpy(nb).to(b.device),
                           torch.from_numpy(np.ascontiguousarray(mask[None, p:e])).to(b.device))
        torch.cuda.synchronize()
        if int(st[0]) != 0:
            raise RuntimeError("decoder status %d" % int(st[0]))
        out[p:e] = pcm.cpu().numpy()[0]
        p = e
    return out.reshape(-1)


def _opt(argv, name, default):
    for i, a in enumerate(argv):
        if a.lower() == name.lower() and i + 1 < len(argv):
            return int(argv[i + 1])
    return default


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) < 3 or argv[0] not in ("enc", "dec"):
        print(__doc__)
        return 2
    mdi = _opt(argv, "-MDI", 0)
    fs = _opt(argv, "-Fs_API", 16000)
    if fs not in (16000, 32000):
        print("-Fs_API: 16000 or 32000")
        return 2
    framesize = _opt(argv, "-framesize", 40)
    if framesize not in (20, 40):
        print("-framesize: 40 or 20")
        return 2
    if argv[0] == "enc":
        recs = encode_pcm(np.fromfile(argv[1], np.int16), rate=_opt(argv, "-rate", 13600), use_md_index=mdi,
                          joint=1 if _opt(argv, "-joint", 0) == 1 else 0, dtx=_opt(argv, "-DTX", 0), samplerate=fs, framesize_ms=framesize)
        open(argv[2], "wb").write(write_bit_container(recs))
        print("%d packets of %d ms, %.3f kbps" % (len(recs), framesize, sum(r[1] for r in recs) * 8 / max(len(recs), 1) / float(framesize)))
    else:
        pcm = decode_records(parse_bit_container(open(argv[1], "rb").read()), loss_perc=_opt(argv, "-loss", 0), use_md_index=mdi,
                             joint=1 if _opt(argv, "-joint", 0) == 1 else 0, samplerate=fs, dec_mode=_opt(argv, "-dec_mode", 0), framesize_ms=framesize)
        pcm.astype(np.int16).tofile(argv[2])
        print("%d packets of %d ms decoded" % (pcm.size // (PACKET_SAMPLES * fs // 16000 * framesize // 40), framesize))
    return 0


if __name__ == "__main__":
    sys.exit(main())
