"""CPU oracle for the batched 1D complex FFT path (test infrastructure, never shipped).

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import this
package; nothing under fft_amd/ does.
"""
from .fft_oracle import (  # noqa: F401
    build,
    dft_direct,
    fft_radix2,
    fft_numpy,
    fft,
    circular_conv,
    circular_conv_fft,
    valid_correlation,
    rel_l2,
    max_abs_err,
)
