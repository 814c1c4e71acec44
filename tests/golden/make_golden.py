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


if __name__ == "__main__":
    main()
