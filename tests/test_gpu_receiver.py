"""the receiver front end (solo_batch_decode_split).  The descriptions of every packet arrive as separate
network packets in two arrival slots -- missing, swapped or duplicated -- and the GPU assembles the decoder call itself.
Checked against the ordinary batched decode with the equivalent arrival mask (which the other GPU tests pin to the compiled
reference) and, for useMDIndex = 1, against the host emulation of the kernel source."""
import numpy as np
import pytest

import refcodec as R
import solo_testlib as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU test run without a GPU"
    return torch


def _scenario(torch, mdi, seed, fs=16000):
    import solo_amd
    N, P, S = 24, 10, 256
    pcm = np.stack([(R.synth_stream if fs == 16000 else T.synth_stream_32k)(300 + i, P) for i in range(N)])
    enc = solo_amd.SoloBatch(N, rate=13600 if fs == 16000 else 24000, encoder=True, decoder=False, slot_bytes=512, use_md_index=mdi, samplerate=fs)
    bits, nb, st = enc.encode(torch.from_numpy(pcm).to(enc.device))
    torch.cuda.synchronize()
    hb, hn = bits.cpu().numpy(), nb.cpu().numpy()
    rng = np.random.default_rng(seed)
    recv = T.bernoulli_recv(N, P, 0.3, seed)
    dA = np.zeros((N, P, S), np.uint8); dB = np.zeros((N, P, S), np.uint8)
    lA = np.zeros((N, P), np.int16); lB = np.zeros((N, P), np.int16)
    for i in range(N):
        for p in range(P):
            n0, n1 = int(hn[i, p, 0]), int(hn[i, p, 1])
            md1, md2 = hb[i, p, :n0 - n1], hb[i, p, n0 - n1:n0]
            a = md1 if recv[i, p] & 1 else None
            b = md2 if recv[i, p] & 2 else None
            if mdi:
                r = rng.integers(0, 3)
                if r == 1:
                    a, b = b, a                                  # swapped arrival slots
                elif r == 2 and a is not None and b is None:
                    b = a                                        # the same description twice
            if a is not None:
                dA[i, p, :a.size] = a; lA[i, p] = a.size
            if b is not None:
                dB[i, p, :b.size] = b; lB[i, p] = b.size
    return hb, hn, recv, dA, lA, dB, lB


@pytest.mark.parametrize("mdi,fs", [(0, 16000), (1, 16000), (1, 32000)])
def test_split_arrivals_vs_compiled_reference_and_masked_decode(torch_cuda, mdi, fs):
    import solo_amd
    torch = torch_cuda
    hb, hn, recv, dA, lA, dB, lB = _scenario(torch, mdi, 11 + mdi, fs)
    N, P = recv.shape
    d1 = solo_amd.SoloBatch(N, encoder=False, decoder=True, slot_bytes=512, use_md_index=mdi, samplerate=fs)
    dev = d1.device
    want, st = d1.decode(torch.from_numpy(hb).to(dev), torch.from_numpy(hn).to(dev), torch.from_numpy(recv).to(dev))
    d2 = solo_amd.SoloBatch(N, encoder=False, decoder=True, slot_bytes=512, use_md_index=mdi, samplerate=fs)
    got, st2 = d2.decode_split(torch.from_numpy(dA).to(dev), torch.from_numpy(lA).to(dev), torch.from_numpy(dB).to(dev),
                               torch.from_numpy(lB).to(dev))
    torch.cuda.synchronize()
    assert int(st.abs().max()) == 0 and int(st2.abs().max()) == 0
    assert np.array_equal(got.cpu().numpy(), want.cpu().numpy())
    assert len({int(m) for m in recv.ravel()}) == 4
    # both against the COMPILED REFERENCE decoder fed the correctly ordered (ptr, nBytes, lostflag) call of test/dec_main.c:255-378
    # (incl. the 32 kHz mode, which tests/test_pinned_corners.py does not cover)
    assert R.have_ref("fix"), "oracle/_ref did not travel with the snapshot"
    w = got.cpu().numpy()
    for i in range(N):
        d = R.RefDecoder("fix", use_md_index=mdi, samplerate=fs)
        for p in range(P):
            n0, n1 = int(hn[i, p, 0]), int(hn[i, p, 1])
            m = int(recv[i, p])
            out, ret = d.decode(*R.map_loss(hb[i, p, :n0].tobytes(), n0, n1, not (m & 1), not (m & 2)))
            assert ret == 0 and np.array_equal(out, w[i, p]), (i, p, m)


def test_oversized_split_packet_is_rejected(torch_cuda):
    import solo_amd
    torch = torch_cuda
    d = solo_amd.SoloBatch(1, encoder=False, decoder=True, slot_bytes=512)
    dA = torch.zeros((1, 1, 400), dtype=torch.uint8, device=d.device)
    lA = torch.full((1, 1), 300, dtype=torch.int16, device=d.device)
    pcm, st = d.decode_split(dA, lA, dA.clone(), torch.zeros_like(lA))
    torch.cuda.synchronize()
    assert int(st[0]) == -11
