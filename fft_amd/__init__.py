"""fft_amd: MI355X-native batched 1D complex FFT behind ThunderFFT's host entry points.

The product is libthunderfft_hip.so (hand-written gfx950 kernels + a C ABI, see
include/thunderfft_c.h). This package is the thin Python host side used by the tests and
bench.py: it binds the C ABI with ctypes and uses torch only for device memory and streams.
"""
from .api import (  # noqa: F401
    TFFT_F32,
    TFFT_F16,
    TFFT_BF16,
    ThunderFFTError,
    ThunderFFTInitialize,
    ThunderFFTFinalize,
    ThunderFFT_global,
    is_supported,
    plan_info,
    set_option,
    exec_options,
    workspace_bytes,
    bench_access_pattern,
    fft,
    SUPPORTED_N,
)
from . import conv, synth  # noqa: F401
