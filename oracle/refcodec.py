"""ctypes driver for the COMPILED, UNMODIFIED reference (oracle/_ref/libsolo_ref_{fix,flp}.so).

TEST INFRASTRUCTURE ONLY.  Nothing in solo_amd/ imports this module.  It is used by tests/,
by __graft_entry__.smoke() and by bench.py's cpu_baseline leg as the checker / reported baseline.

The library is built by `make -C oracle ref` from the sources where they lie under
/root/reference (see oracle/Makefile); the built .so travels to the GPU box, the sources do not.

Interface driven (reference file:line):
  AGR_Sate_Encoder_Init / _Encode / _Uninit   JC1_SDK_SRC_ARM/interface/AGR_JC1_SDK_API.h:33-47
  AGR_Sate_Decoder_Init / _Decode / _Uninit   JC1_SDK_SRC_ARM/interface/AGR_JC1_SDK_API.h:49-64
Harness behaviour mirrored (framing, loss -> lostflag mapping):
  JC1_SDK_SRC_ARM/test/enc_main.c:190-253, JC1_SDK_SRC_ARM/test/dec_main.c:202-378
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PACKET_SAMPLES = 640
MAX_FRAME_BYTES = 1024


class USER_Ctrl_enc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "mode", "targetRate_bps", "samplerate", "dtx_enable", "framesize_ms",
        "joint_enable", "joint_mode", "useMDIndex")]


class USER_Ctrl_dec(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "packetLoss_perc", "samplerate", "framesize_ms", "joint_enable", "joint_mode", "useMDIndex")]


def ref_lib_path(kind="fix"):
    return os.path.join(_HERE, "_ref", "libsolo_ref_%s.so" % kind)


def have_ref(kind="fix"):
    return os.path.exists(ref_lib_path(kind))


_libs = {}


def load_ref(kind="fix"):
    if kind not in _libs:
        lib = C.CDLL(ref_lib_path(kind))
        lib.AGR_Sate_Encoder_Init.restype = C.c_void_p
        lib.AGR_Sate_Encoder_Init.argtypes = [C.POINTER(USER_Ctrl_enc)]
        lib.AGR_Sate_Encoder_Encode.restype = C.c_int32
        lib.AGR_Sate_Encoder_Encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        lib.AGR_Sate_Encoder_Uninit.argtypes = [C.c_void_p]
        lib.AGR_Sate_Decoder_Init.restype = C.c_void_p
        lib.AGR_Sate_Decoder_Init.argtypes = [C.POINTER(USER_Ctrl_dec)]
        lib.AGR_Sate_Decoder_Decode.restype = C.c_int32
        lib.AGR_Sate_Decoder_Decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        lib.AGR_Sate_Decoder_Uninit.argtypes = [C.c_void_p]
        _libs[kind] = lib
    return _libs[kind]


def default_enc_ctrl(rate=13600, use_md_index=0, joint=0, dtx=0, samplerate=16000, framesize_ms=40):
    # defaults of the reference CLI: JC1_SDK_SRC_ARM/test/enc_main.c:92-99; joint=1: `-joint 1` (40 ms high-band frame);
    # samplerate=32000: `-Fs_API 32000` (16 kHz bands, SILK wide band)
    return USER_Ctrl_enc(mode=2, targetRate_bps=rate, samplerate=samplerate, dtx_enable=1 if dtx else 0,
                         framesize_ms=framesize_ms, joint_enable=1 if joint else 0, joint_mode=1 if joint else 0, useMDIndex=use_md_index)


def default_dec_ctrl(use_md_index=0, joint=0, samplerate=16000, framesize_ms=40):
    return USER_Ctrl_dec(packetLoss_perc=0, samplerate=samplerate, framesize_ms=framesize_ms,
                         joint_enable=1 if joint else 0, joint_mode=1 if joint else 0, useMDIndex=use_md_index)


class RefEncoder:
    def __init__(self, kind="fix", rate=13600, joint=0, dtx=0, samplerate=16000, use_md_index=0, framesize_ms=40):
        # use_md_index = 1: every description starts with its range-coded index (SKP_Silk_encode_parameters.c:50-51)
        self.lib = load_ref(kind)
        self.ctrl = default_enc_ctrl(rate, use_md_index=use_md_index, joint=joint, dtx=dtx, samplerate=samplerate, framesize_ms=framesize_ms)
        self.packet_samples = PACKET_SAMPLES * samplerate // 16000 * framesize_ms // 40
        self.h = self.lib.AGR_Sate_Encoder_Init(C.byref(self.ctrl))
        assert self.h
        self._bits = np.zeros(MAX_FRAME_BYTES, np.uint8)
        self._nb = np.zeros(6, np.int16)

    def encode(self, pcm640):
        """-> (payload bytes, nBytes0 total, nBytes1 = len(MD2)+HB)"""
        pcm = np.ascontiguousarray(pcm640, dtype=np.int16)
        assert pcm.size == self.packet_samples
        self._nb[:] = 0
        n = self.lib.AGR_Sate_Encoder_Encode(self.h, pcm.ctypes.data, self._bits.ctypes.data,
                                             MAX_FRAME_BYTES, self._nb.ctypes.data)
        return self._bits[:n].tobytes(), int(self._nb[0]), int(self._nb[1])

    def close(self):
        if self.h:
            self.lib.AGR_Sate_Encoder_Uninit(self.h)
            self.h = None

    __del__ = close


class RefDecoder:
    def __init__(self, kind="fix", joint=0, samplerate=16000, use_md_index=0, framesize_ms=40):
        # use_md_index = 1: the decoder reads the description index first (SKP_Silk_decode_parameters.c:55-57)
        self.lib = load_ref(kind)
        self.ctrl = default_dec_ctrl(use_md_index=use_md_index, joint=joint, samplerate=samplerate, framesize_ms=framesize_ms)
        self.packet_samples = PACKET_SAMPLES * samplerate // 16000 * framesize_ms // 40
        self.h = self.lib.AGR_Sate_Decoder_Init(C.byref(self.ctrl))
        assert self.h
        self._pcm = np.zeros(1920, np.int16)
        self._ns = np.zeros(1, np.int16)
        self._nb = np.zeros(6, np.int16)

    def decode(self, payload, nbytes0, nbytes1, lostflag):
        """payload: bytes handed to the decoder (already offset per the harness mapping)."""
        buf = np.zeros(MAX_FRAME_BYTES, np.uint8)
        pl = np.frombuffer(payload, np.uint8)
        buf[:pl.size] = pl
        self._nb[:] = 0
        self._nb[0] = nbytes0
        self._nb[1] = nbytes1
        ret = self.lib.AGR_Sate_Decoder_Decode(self.h, self._pcm.ctypes.data, self._ns.ctypes.data,
                                               buf.ctypes.data, self._nb.ctypes.data, int(lostflag))
        # what the call left in the caller's nBytes[0..1] (AGR_BWE_decode_frame_FIX.c:150-169) and *nSamplesOut
        self.nbytes_after = (int(self._nb[0]), int(self._nb[1]))
        self.nsamples_out = int(self._ns[0])
        return self._pcm[:self.packet_samples].copy(), ret

    def close(self):
        if self.h:
            self.lib.AGR_Sate_Decoder_Uninit(self.h)
            self.h = None

    __del__ = close


# ---------------------------------------------------------------------------------------------
# harness-side helpers shared by the tests (pure data plumbing, no codec arithmetic)
# ---------------------------------------------------------------------------------------------
def map_loss(payload, n0, n1, lost_md1, lost_md2):
    """(payload, nBytes0, nBytes1, lostflag) the decoder is called with for a given loss pattern.
    Mirrors JC1_SDK_SRC_ARM/test/dec_main.c:255-378 (_SIMU1_ mapping; both branches are identical
    in effect for the 2-description case)."""
    if not lost_md1 and not lost_md2:
        return payload, n0, n1, 4
    if not lost_md1 and lost_md2:
        return payload[:n0 - n1], n0 - n1, 0, 2
    if lost_md1 and not lost_md2:
        return payload[n0 - n1:], n1, 0, 3
    return b"", n0, n1, 1


def skp_rand(seed):
    # JC1_SDK_SRC_ARM/test/dec_main.c:24 (int32 wrap-around)
    return (907633515 + (seed * 196314165)) & 0xFFFFFFFF


def cli_loss_pattern(n_packets, loss_perc, md_nbytes=None):
    """The loss draws of the reference decoder CLI (`-loss P`): rand_seed=1, two draws on every
    EVEN packet (run_count % 2 == 0), pattern reused for the following odd packet
    (JC1_SDK_SRC_ARM/test/dec_main.c:236-252)."""
    seed = 1
    out = []
    lost = [0, 0]
    thr = np.float32(loss_perc) / np.float32(100.0)
    for p in range(n_packets):
        if p % 2 == 0:
            for j in range(2):
                seed = skp_rand(seed)
                s = seed - (1 << 32) if seed & 0x80000000 else seed
                v = np.float32((s >> 16) + (1 << 15)) / np.float32(65535.0)
                if v >= thr:
                    lost[j] = 1 if (md_nbytes is not None and md_nbytes[p][j] == 0) else 0
                else:
                    lost[j] = 1
        out.append(tuple(lost))
    return out


from solo_amd.synth import synth_stream  # noqa: E402,F401  (the workload generator lives with the host package)
