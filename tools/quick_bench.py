#!/usr/bin/env python3
"""Quick A/B timing of encode / decode for a given build of the library (SOLO_LIB_OVERRIDE), with a parity check
against the committed goldens first.   SOLO_LIB_OVERRIDE=build/x.so python tools/quick_bench.py [streams] [packets] [16000|32000]
(32000: the wide-band build, 1280-sample packets at 24 kbps)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, solo_amd
from solo_amd.synth import synth_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
P = int(sys.argv[2]) if len(sys.argv) > 2 else 10
FS = int(sys.argv[3]) if len(sys.argv) > 3 else 16000
if FS == 16000:
    z = np.load(os.path.join(ROOT, "tests/golden/synth8x25.npz"))
    b = solo_amd.SoloBatch(8, encoder=True, decoder=True, slot_bytes=512)
    bits, nb, st = b.encode(torch.from_numpy(z["pcm"]).cuda())
    ok = np.array_equal(nb.cpu().numpy(), z["nbytes"]) and all(
        np.array_equal(bits[i, p, :int(nb[i, p, 0])].cpu().numpy(), z["bits"][i, p, :int(nb[i, p, 0])]) for i in range(8) for p in range(25))
    pcm, st2 = b.decode(bits, nb, torch.from_numpy(z["recv"]).cuda())
    ok2 = np.array_equal(pcm.cpu().numpy(), z["dec_loss"])
    b = solo_amd.SoloBatch(N, encoder=True, decoder=True, slot_bytes=512)
    x = torch.from_numpy(synth_batch(0, N, P, workers=16)).cuda()
else:
    z = np.load(os.path.join(ROOT, "tests/golden/wb4x20.npz"))
    b = solo_amd.SoloBatch(3, rate=24000, encoder=True, decoder=True, slot_bytes=512, samplerate=32000)
    bits, nb, st = b.encode(torch.from_numpy(z["pcm"][:3].copy()).cuda())
    ok = np.array_equal(nb.cpu().numpy(), z["nbytes"][:3]) and all(
        np.array_equal(bits[i, p, :int(nb[i, p, 0])].cpu().numpy(), z["bits"][i, p, :int(nb[i, p, 0])]) for i in range(3) for p in range(20))
    pcm, st2 = b.decode(bits, nb, torch.from_numpy(z["recv"][:3].copy()).cuda())
    ok2 = np.array_equal(pcm.cpu().numpy(), z["dec_loss"][:3])
    b = solo_amd.SoloBatch(N, rate=24000, encoder=True, decoder=True, slot_bytes=512, samplerate=32000)
    x = torch.from_numpy(synth_batch(0, N, 2 * P, workers=16).reshape(N, P, 1280)).cuda()
bits, nb, st = b.encode(x); out, st2 = b.decode(bits, nb); torch.cuda.synchronize()
LOSS = float(os.environ.get("LOSS", "0"))          # probability that a description is lost (BASELINE config 4: 0.3), decode leg only
recv = None
if LOSS > 0:
    rng = np.random.default_rng(4242)
    m = (rng.random((N, P)) >= LOSS).astype(np.uint8) | ((rng.random((N, P)) >= LOSS).astype(np.uint8) << 1)
    m[:, 0] = 3
    recv = torch.from_numpy(m).cuda()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
R = 3
te = td = 0.0
for _ in range(R):
    ev[0].record(); b.encode(x, bits, nb, st); ev[1].record(); b.decode(bits, nb, recv, out, st2); ev[2].record(); torch.cuda.synchronize()
    te += ev[0].elapsed_time(ev[1]); td += ev[1].elapsed_time(ev[2])
print("%s parity enc=%s dec=%s | encode %.2f ms (%.0f pkt/s) decode %.2f ms (%.0f pkt/s) round trip %.0f pkt/s" % (
    os.environ.get("SOLO_LIB_OVERRIDE", "default"), ok, ok2, te / R, N * P * R / te * 1e3, td / R, N * P * R / td * 1e3, N * P * R / (te + td) * 1e3))
try:
    b.set_timing(True)
    b.encode(x, bits, nb, st); b.decode(bits, nb, None, out, st2); torch.cuda.synchronize()
    print("  kernels (ms): " + "  ".join("%s %.2f" % kv for kv in b.last_kernel_ms().items()))
except Exception as e:      # older builds without the timing entry points
    print("  (no per-kernel timing: %s)" % e)
