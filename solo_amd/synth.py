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
