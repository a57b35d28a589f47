#!/usr/bin/env python3
"""Cost of the six legacy entry points (AGR_Sate_*: ONE stream per handle, host buffers, every call = host-to-device copy + a
four-kernel pipeline + device-to-host copies) next to the compiled reference doing the same calls on one host core: the reference's
own CLI mains linked with this library (oracle/_ref/JC1*_solo) and with the reference (JC1*_ref) encode / decode the reference's
speech sample (191 packets of 40 ms), wall time of the whole process minus an empty run's start-up, and the per-call time measured
in-process through ctypes.    python tools/legacy_api_cost.py   (GPU box; prints one JSON line)"""
import ctypes as C, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
REF = os.path.join(ROOT, "oracle", "_ref")
PCM = os.path.join(ROOT, "tests", "golden", "Ch_f1_raw.pcm")


def wall(cmd, n=3):
    best = 1e9
    for _ in range(n):
        t = time.perf_counter()
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        best = min(best, time.perf_counter() - t)
    return best


def main():
    out = {}
    tmp = tempfile.mkdtemp()
    for tag in ("solo", "ref"):
        enc, dec = os.path.join(REF, "JC1Encoder_" + tag), os.path.join(REF, "JC1Decoder_" + tag)
        bit, pcm = os.path.join(tmp, tag + ".bit"), os.path.join(tmp, tag + ".pcm")
        out["cli_encode_s_" + tag] = round(wall([enc, PCM, bit, "-mode", "2", "-Fs_API", "16000", "-rate", "13600"]), 4)
        out["cli_decode_s_" + tag] = round(wall([dec, bit, pcm, "-Fs_API", "16000"]), 4)
    # in-process, per call (handle creation excluded)
    import solo_amd, refcodec as R
    lib = solo_amd.load_library()
    x = np.fromfile(PCM, np.int16)
    P = x.size // 640
    ctrl = solo_amd.default_enc_ctrl()
    h = lib.AGR_Sate_Encoder_Init(C.byref(ctrl))
    buf = np.zeros(1100, np.uint8); nb = np.zeros(6, np.int16)
    recs = []
    lib.AGR_Sate_Encoder_Encode(h, x[:640].ctypes.data, buf.ctypes.data, 1024, nb.ctypes.data)      # warm-up (allocations inside the handle)
    lib.AGR_Sate_Encoder_Uninit(h)
    h = lib.AGR_Sate_Encoder_Init(C.byref(ctrl))
    t = time.perf_counter()
    for p in range(P):
        xp = np.ascontiguousarray(x[p * 640:(p + 1) * 640])
        n = lib.AGR_Sate_Encoder_Encode(h, xp.ctypes.data, buf.ctypes.data, 1024, nb.ctypes.data)
        recs.append((buf[:n].tobytes(), int(nb[0]), int(nb[1])))
    out["encode_ms_per_call_solo"] = round((time.perf_counter() - t) / P * 1e3, 4)
    lib.AGR_Sate_Encoder_Uninit(h)
    dctrl = solo_amd.default_dec_ctrl()
    hd = lib.AGR_Sate_Decoder_Init(C.byref(dctrl))
    o = np.zeros(1920, np.int16); ns = np.zeros(1, np.int16)
    t = time.perf_counter()
    for pl, n0, n1 in recs:
        b = np.zeros(1100, np.uint8); b[:len(pl)] = np.frombuffer(pl, np.uint8)
        nbv = np.array([n0, n1, 0, 0, 0, 0], np.int16)
        lib.AGR_Sate_Decoder_Decode(hd, o.ctypes.data, ns.ctypes.data, b.ctypes.data, nbv.ctypes.data, 4)
    out["decode_ms_per_call_solo"] = round((time.perf_counter() - t) / P * 1e3, 4)
    lib.AGR_Sate_Decoder_Uninit(hd)
    e = R.RefEncoder("fix"); d = R.RefDecoder("fix")
    t = time.perf_counter()
    for p in range(P):
        e.encode(x[p * 640:(p + 1) * 640])
    out["encode_ms_per_call_ref"] = round((time.perf_counter() - t) / P * 1e3, 4)
    t = time.perf_counter()
    for pl, n0, n1 in recs:
        d.decode(pl, n0, n1, 4)
    out["decode_ms_per_call_ref"] = round((time.perf_counter() - t) / P * 1e3, 4)
    out["packets"] = P
    print(json.dumps(out))


if __name__ == "__main__":
    main()
