"""Python face of the CPU oracle (oracle/dft_oracle.c) plus a numpy.fft cross-check.

Test infrastructure only. Every function takes `x` of shape [batch, n] (or [n]) and returns
complex128 of the same shape: forward = sum_n x[n] exp(-2 pi i n k / N), inverse = the "+"
sign, both unnormalised -- the convention the reference inherits from rotate()
(thunderfft/include/thunderfft/detail/util_arith.h:85-91) and its CUFFT_FORWARD comparator
(thunderfft/sample/bench/baseline/baseline_cufft.cu:46-47).

The reference holds no golden vectors for this path, so parity is pinned by this definition
and the closed-form answers of the reference's own test inputs (see dft_oracle.c header).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_dft.so")
_lib = None


def build(force=False):
    """Compile dft_oracle.c with gcc (a few hundred ms). Returns the path of the .so."""
    src = os.path.join(_HERE, "dft_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", _SO, src, "-lm"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        lib = ctypes.CDLL(_SO)
        for name in ("oracle_dft_direct", "oracle_fft_radix2"):
            fn = getattr(lib, name)
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_int]
        lib.oracle_circular_conv.restype = ctypes.c_int
        lib.oracle_circular_conv.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_longlong, ctypes.c_int]
        _lib = lib
    return _lib


def _run(name, x, forward):
    x = np.asarray(x)
    shape = x.shape
    x2 = np.ascontiguousarray(x.reshape(-1, shape[-1]).astype(np.complex128))
    batch, n = x2.shape
    out = np.empty_like(x2)
    rc = getattr(_load(), name)(x2.ctypes.data, out.ctypes.data, n, batch, -1 if forward else 1)
    if rc != 0:
        raise ValueError(f"{name} failed with code {rc} (n={n}, batch={batch})")
    return out.reshape(shape)


def dft_direct(x, forward=True):
    """O(N^2) fp64 DFT, angle index reduced mod N in integers."""
    return _run("oracle_dft_direct", x, forward)


def fft_radix2(x, forward=True):
    """fp64 radix-2 decimation-in-time FFT; n must be a power of two."""
    return _run("oracle_fft_radix2", x, forward)


def fft_numpy(x, forward=True):
    """numpy.fft on complex128 (pocketfft): the independent cross-check."""
    x = np.asarray(x).astype(np.complex128)
    if forward:
        return np.fft.fft(x, axis=-1)
    return np.fft.ifft(x, axis=-1) * x.shape[-1]


def fft(x, forward=True):
    """The oracle the GPU path is compared with."""
    return fft_radix2(x, forward)


def circular_conv(x, h):
    """y[s] = x[s] (*) h[s % D], circular, by the O(N K) definition in fp64 -- the reference's own
    validation loop (thunderfft/sample/conv1d/conv1d_fullfft_l4096_thunder.cu:472-483).
    x: [signals, N] complex, h: [D, K] complex; returns complex128 [signals, N]."""
    x2 = np.ascontiguousarray(np.asarray(x).astype(np.complex128))
    h2 = np.ascontiguousarray(np.asarray(h).astype(np.complex128))
    signals, n = x2.shape
    D, K = h2.shape
    y = np.empty_like(x2)
    rc = _load().oracle_circular_conv(x2.ctypes.data, h2.ctypes.data, y.ctypes.data, n, K, signals, D)
    if rc != 0:
        raise ValueError(f"oracle_circular_conv failed with code {rc}")
    return y


def circular_conv_fft(x, h):
    """The same result through the oracle's own transforms: IDFT(DFT(x) * DFT(h padded)) / N."""
    x2 = np.asarray(x).astype(np.complex128)
    h2 = np.asarray(h).astype(np.complex128)
    signals, n = x2.shape
    D, K = h2.shape
    hp = np.zeros((D, n), dtype=np.complex128)
    hp[:, :K] = h2
    H = fft_radix2(hp, True)
    X = fft_radix2(x2, True)
    return fft_radix2(X * H[np.arange(signals) % D], False) / n


def valid_correlation(x, taps):
    """out[i] = sum_k x[i + k] * taps[k], i < len(x) - len(taps) + 1, no conjugation: the reference's
    reference_conv1d_valid (thunderfft/sample/conv1d/main.cu:156-174) in fp64."""
    x = np.asarray(x).astype(np.complex128)
    taps = np.asarray(taps).astype(np.complex128)
    n_out = x.shape[0] - taps.shape[0] + 1
    out = np.zeros(n_out, dtype=np.complex128)
    for k in range(taps.shape[0]):
        out += x[k:k + n_out] * taps[k]
    return out


def rel_l2(test, ref):
    """Relative L2 error over the whole batch, the reference's l2_rel
    (thunderfft/sample/fft1d/main.cu:48-58) in fp64."""
    test = np.asarray(test).astype(np.complex128)
    ref = np.asarray(ref).astype(np.complex128)
    num = np.sum(np.abs(test - ref) ** 2)
    den = np.sum(np.abs(ref) ** 2)
    return float(np.sqrt(num / den)) if den > 0 else float(np.sqrt(num))


def max_abs_err(test, ref):
    """Largest complex modulus of the difference, the reference's check_max_abs_err
    (thunderfft/sample/bench/util/helper.h:94-120); inf if any output is not finite."""
    test = np.asarray(test).astype(np.complex128)
    ref = np.asarray(ref).astype(np.complex128)
    if not np.all(np.isfinite(test.view(np.float64))):
        return float("inf")
    return float(np.max(np.abs(test - ref))) if test.size else 0.0
