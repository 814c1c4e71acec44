"""Writes tests/golden/fft_golden.npz: seeded inputs and their fp64 transforms.

The reference holds no golden vectors (it checks against cuFFT at run time only), so these
fixtures are produced from the definition with numpy.fft on complex128 and cross-checked
against the direct O(N^2) sum before they are written. Inputs are stored as complex64 (what
the device path consumes); expected outputs as complex128 of the complex64 inputs.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fft_amd import synth  # noqa: E402


def direct_dft(x, sign):
    n = x.shape[-1]
    k = np.arange(n)
    w = np.exp(sign * 2j * np.pi * ((k[:, None] * k[None, :]) % n) / n)
    return x.astype(np.complex128) @ w.T


def main():
    out = {}
    for n, batch in ((64, 5), (128, 3), (256, 3), (512, 2), (1024, 2), (2048, 1), (4096, 1), (8192, 1), (16384, 1)):
        x = synth.uniform_complex(batch, n, seed=1000 + n)
        fwd = np.fft.fft(x.astype(np.complex128), axis=-1)
        inv = np.fft.ifft(x.astype(np.complex128), axis=-1) * n
        if n <= 1024:
            assert np.max(np.abs(direct_dft(x, -1) - fwd)) < 1e-9 * n
            assert np.max(np.abs(direct_dft(x, +1) - inv)) < 1e-9 * n
        out[f"x_{n}"] = x
        out[f"fwd_{n}"] = fwd
        out[f"inv_{n}"] = inv
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fft_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    conv_golden()


def conv_golden():
    """tests/golden/conv_golden.npz: the two convolution forms by their definitions, plain numpy sums in fp64.
    Circular convolution of length 4096 (the reference's validation loop,
    thunderfft/sample/conv1d/conv1d_fullfft_l4096_thunder.cu:472-483): y[b][d][n] = sum_k x[b][d][(n - k) mod N]
    h[d][k]. Valid correlation of overlap-save tiles (sample/conv1d/main.cu:156-174): out[i] = sum_k x[i + k] t[k].
    Inputs are values fp16 represents exactly (multiples of 2^-8 below 1), so that every storage type sees the
    same numbers."""
    rng = np.random.default_rng(4096)
    n, B, D, K = 4096, 2, 3, 9
    q = lambda shape: rng.integers(0, 256, size=shape) / 256.0
    x = q((B, D, n)) + 1j * q((B, D, n))
    xr = q((B, D, n))
    h = (rng.integers(0, 16, size=(D, K)) + 1j * rng.integers(0, 16, size=(D, K))) / 16.0
    hr = rng.integers(0, 16, size=(D, K)) / 16.0
    idx = (np.arange(n)[:, None] - np.arange(K)[None, :]) % n  # [n, K]

    def circ(sig, taps):
        y = np.zeros(sig.shape, dtype=np.complex128)
        for d in range(D):
            y[:, d, :] = (sig[:, d, :][:, idx] * taps[d][None, None, :]).sum(axis=-1)
        return y

    L, KT = 1000, 9
    xt = q(L) + 1j * q(L)
    taps = (rng.integers(0, 16, size=KT) + 1j * rng.integers(0, 16, size=KT)) / 16.0
    valid = np.array([np.dot(xt[i:i + KT], taps) for i in range(L - KT + 1)])
    out = {"x": x, "h": h, "y": circ(x, h), "x_real": xr, "h_real": hr, "y_real": circ(xr + 0j, hr + 0j),
           "tiles_x": xt, "tiles_taps": taps, "tiles_y": valid}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conv_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
