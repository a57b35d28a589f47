"""Synthetic 16 kHz wide-band workload generator: used by bench.py and the tests."""
import numpy as np

PACKET_SAMPLES = 640


def synth_stream(stream_index, n_packets, base_seed=0x50100000):
    """Synthetic speech-like 16 kHz int16 stream (BASELINE.md section 4 generator)."""
    rng = np.random.default_rng(base_seed + int(stream_index))
    n = n_packets * PACKET_SAMPLES
    fs = 16000.0
    t = np.arange(n) / fs
    f0 = rng.uniform(90.0, 250.0)
    vib = 1.0 + 0.02 * np.sin(2 * np.pi * 3.0 * t + rng.uniform(0, 2 * np.pi))
    phase = 2 * np.pi * np.cumsum(f0 * vib) / fs
    sig = np.zeros(n)
    h = 1
    while h * f0 * 1.02 < 7500.0:
        sig += np.sin(h * phase + rng.uniform(0, 2 * np.pi)) / h
        h += 1
    # syllabic on/off envelope, ~60 % active, 5 ms ramps
    env = np.zeros(n)
    pos = 0
    while pos < n:
        on = int(rng.uniform(0.12, 0.45) * fs)
        off = int(rng.uniform(0.05, 0.30) * fs)
        env[pos:pos + on] = rng.uniform(0.3, 1.0)
        pos += on + off
    k = int(0.005 * fs)
    env = np.convolve(env, np.ones(k) / k, mode="same")
    sig = sig * env
    peak = np.max(np.abs(sig)) + 1e-9
    sig = sig / peak * 12000.0 + rng.normal(0.0, 200.0, n)
    return np.clip(np.rint(sig), -32768, 32767).astype(np.int16).reshape(n_packets, PACKET_SAMPLES)


def synth_batch(first_stream, n_streams, n_packets, workers=8):
    """int16 [n_streams, n_packets, 640]; stream i is synth_stream(first_stream + i, n_packets)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=workers) as ex:
        rows = list(ex.map(lambda i: synth_stream(first_stream + i, n_packets), range(n_streams)))
    return np.stack(rows)


EDGE_FAMILIES = 14


def edge_stream(seed, n_packets):
    """Deliberately un-speech-like 16 kHz int16 stream for parity sweeps (family = seed % EDGE_FAMILIES, the rest of the seed
    varies it): silence with stray LSBs, full-scale noise / square waves / sweeps, DC, sparse impulses, level ramps over 90 dB,
    clipped and very quiet speech-like signals, on / off bursts, high-band-only tones, the Nyquist pattern, sub-audio sines,
    clipped random walks.  They reach the saturation, zero-energy and shift-normalisation branches of the fixed-point code that
    speech-like input rarely takes."""
    seed = int(seed)
    fam = seed % EDGE_FAMILIES
    rng = np.random.default_rng(0x0ED6E000 + seed)
    n = n_packets * PACKET_SAMPLES
    fs = 16000.0
    t = np.arange(n) / fs
    if fam == 0:
        x = np.zeros(n)
        idx = rng.integers(0, n, max(1, n // 500))
        x[idx] = rng.choice([-1.0, 1.0], idx.size)
    elif fam == 1:
        x = rng.integers(-32768, 32768, n).astype(np.float64)
    elif fam == 2:
        x = 32767.0 * np.sign(np.sin(2 * np.pi * rng.uniform(60.0, 900.0) * t + 0.1))
    elif fam == 3:
        x = rng.choice([-1.0, 1.0]) * rng.uniform(8000.0, 32000.0) + rng.normal(0.0, rng.uniform(0.5, 30.0), n)
    elif fam == 4:
        x = np.zeros(n)
        idx = rng.integers(0, n, max(1, n // int(rng.integers(40, 400))))
        x[idx] = rng.choice([-32768.0, 32767.0, 12000.0, -300.0], idx.size)
    elif fam == 5:
        g = 10.0 ** (np.linspace(-90.0, 0.0, n) / 20.0)
        if rng.random() < 0.5:
            g = g[::-1]
        x = 32767.0 * g * np.sin(2 * np.pi * np.cumsum(rng.uniform(100.0, 3000.0) * (1.0 + 0.3 * np.sin(2 * np.pi * 1.5 * t))) / fs)
    elif fam == 6:
        x = synth_stream(seed, n_packets).reshape(-1).astype(np.float64) * rng.uniform(3.0, 12.0)
    elif fam == 7:
        x = synth_stream(seed, n_packets).reshape(-1).astype(np.float64) / rng.uniform(64.0, 2048.0)
    elif fam == 8:
        x = synth_stream(seed, n_packets).reshape(-1).astype(np.float64) * 2.5
        pos = 0
        while pos < n:
            on = int(rng.integers(40, 2000)); off = int(rng.integers(40, 3000))
            x[pos + on:pos + on + off] = 0.0
            pos += on + off
    elif fam == 9:
        f = np.exp(np.linspace(np.log(50.0), np.log(7900.0), n))
        x = 32000.0 * np.sin(2 * np.pi * np.cumsum(f) / fs)
    elif fam == 10:
        x = 15000.0 * np.sin(2 * np.pi * rng.uniform(5000.0, 6500.0) * t) + 15000.0 * np.sin(2 * np.pi * rng.uniform(6600.0, 7900.0) * t + 1.0)
    elif fam == 11:
        x = 32767.0 * np.where(np.arange(n) % 2 == 0, 1.0, -1.0) * (1.0 if rng.random() < 0.5 else np.sign(np.sin(2 * np.pi * 7.0 * t) + 0.5))
    elif fam == 12:
        x = rng.uniform(10000.0, 32767.0) * np.sin(2 * np.pi * rng.uniform(5.0, 45.0) * t)
    else:
        x = np.cumsum(rng.normal(0.0, rng.uniform(50.0, 3000.0), n))
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16).reshape(n_packets, PACKET_SAMPLES)


def edge_batch(first_seed, n_streams, n_packets):
    """int16 [n_streams, n_packets, 640]; stream i is edge_stream(first_seed + i, n_packets)."""
    return np.stack([edge_stream(first_seed + i, n_packets) for i in range(n_streams)])
