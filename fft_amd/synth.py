"""Synthetic inputs: a counter-based hash RNG with the same arithmetic as bench/tfft_bench.cpp,
and the reference's own test signals."""
import numpy as np


def hash_u32(idx, seed=0):
    """lowbias32-style integer hash of (idx, seed); idx is a uint64 array of element counters."""
    x = (idx.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed) * np.uint64(0xD1B54A32D192ED03)) >> np.uint64(32)
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x7FEB352D)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x * np.uint32(0x846CA68B)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def uniform_complex(batch, n, seed=0, scale=1.0):
    """complex64 [batch, n]; real and imaginary parts i.i.d. uniform(-1, 1) * scale."""
    cnt = np.arange(1, 2 * batch * n + 1, dtype=np.uint64)  # counters start at 1
    with np.errstate(over="ignore"):
        u = hash_u32(cnt, seed)
    f = (u >> np.uint32(8)).astype(np.float32) * np.float32(2.0 / (1 << 24)) - np.float32(1.0)
    f = f * np.float32(scale)
    return f.reshape(batch, n, 2).view(np.complex64).reshape(batch, n)


def bench_sine(batch, n):
    """The reference bench input x[i] = sin(2 pi (i mod N) / N) / sqrt(N)
    (thunderfft/sample/bench/main.cu:35-41). Its spectrum is X[1] = -i sqrt(N)/2,
    X[N-1] = +i sqrt(N)/2, zero elsewhere."""
    i = np.arange(n, dtype=np.float64)
    row = (np.sin(2 * np.pi * i / n) / np.sqrt(n)).astype(np.float32)
    return np.broadcast_to(row.astype(np.complex64), (batch, n)).copy()


def fft1d_tones(batch, n):
    """The reference fft1d sample input (thunderfft/sample/fft1d/main.cu:16-30):
    amp_b e^{2 pi i (1+b) t} + 0.1 e^{2 pi i 7 t}, amp_b = 0.5 + 0.1 b, t = i/N.
    Spectrum: X[(1+b) mod N] += N amp_b, X[7] += 0.1 N."""
    b = np.arange(batch, dtype=np.float64)[:, None]
    t = np.arange(n, dtype=np.float64)[None, :] / n
    amp = 0.5 + 0.1 * b
    x = amp * np.exp(2j * np.pi * (1 + b) * t) + 0.1 * np.exp(2j * np.pi * 7 * t)
    return x.astype(np.complex64)
