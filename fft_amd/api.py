"""Host-side mirror of the reference's three entry points, above the C ABI.

Reference (thunderfft/include/thunderfft/thunderfft.cuh):
    template <typename T> void ThunderFFTInitialize(int N);                      // :39-40
    template <typename T> void ThunderFFTFinalize();                             // :43-44
    template <typename T, int N, bool forward = true>
    void ThunderFFT_global(vec2_t<T>* d_input, vec2_t<T>* d_output, int batch,
                           cudaStream_t stream = 0);                             // :98-102
The template parameters T, N and forward become ordinary arguments; tensors stand in for the
raw device pointers. Errors raise ThunderFFTError (the reference throws std::runtime_error
from its twiddle path, detail/util_cuda.h:20-24).
"""
import torch

from . import _lib

TFFT_F32, TFFT_F16, TFFT_BF16 = 0, 1, 2
SUPPORTED_N = (64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384)

_DTYPE_CODE = {
    torch.float32: TFFT_F32,
    torch.complex64: TFFT_F32,
    torch.float16: TFFT_F16,
    torch.bfloat16: TFFT_BF16,
}


class ThunderFFTError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"[ThunderFFT] {message} (code {code})")
        self.code = code


def _check(rc):
    if rc != 0:
        raise ThunderFFTError(rc, _lib.load().tfft_last_error().decode())


def dtype_code(dtype):
    """torch dtype (or 'fp32' / 'fp16' / 'bf16', or an int code) -> TFFT_* code."""
    if isinstance(dtype, int):
        return dtype
    if isinstance(dtype, str):
        return {"fp32": TFFT_F32, "float": TFFT_F32, "fp16": TFFT_F16, "half": TFFT_F16, "bf16": TFFT_BF16}[dtype]
    return _DTYPE_CODE[dtype]


def ThunderFFTInitialize(dtype, N):
    """Build and upload the twiddle table for length N on the current device."""
    _check(_lib.load().tfft_initialize(dtype_code(dtype), int(N)))


def ThunderFFTFinalize(dtype):
    """Free every table of this element type."""
    _check(_lib.load().tfft_finalize(dtype_code(dtype)))


def is_supported(dtype, N, forward=True):
    return bool(_lib.load().tfft_is_supported(dtype_code(dtype), int(N), int(bool(forward))))


def plan_info(dtype, N):
    info = _lib.PlanInfo()
    _check(_lib.load().tfft_get_plan_info(dtype_code(dtype), int(N), info))
    return {
        "n": info.n,
        "threads_per_block": info.threads_per_block,
        "transforms_per_block": info.transforms_per_block,
        "points_per_thread": info.points_per_thread,
        "radix": [r for r in info.radix if r > 1],
        "lds_bytes": info.lds_bytes,
        "blocks_per_cu": info.blocks_per_cu,
    }


def set_option(key, value):
    """Process-wide switch; the only key is "sync" (block after every launch like the reference)."""
    _check(_lib.load().tfft_set_option(key.encode(), int(value)))


def exec_options(options):
    """dict of per-call launch options -> a filled tfft_exec_options (None stays None = defaults).

    Keys: plan, xcd_run_log2, grid_blocks_per_cu, grid_blocks, extra_lds, workspace (a device tensor used as the
    scratch of a multi-pass transform)."""
    if not options:
        return None
    o = _lib.ExecOptions()
    _lib.load().tfft_exec_options_init(o)
    for key, value in options.items():
        if key == "workspace":
            if value is not None:
                o.workspace = value.data_ptr()
                o.workspace_bytes = value.numel() * value.element_size()
        elif key in ("plan", "xcd_run_log2", "grid_blocks_per_cu", "grid_blocks", "extra_lds"):
            setattr(o, key, int(value))
        else:
            raise ThunderFFTError(3, f"unknown launch option '{key}'")
    return o


def workspace_bytes(dtype, N, batch=1):
    """(minimum, best) scratch bytes of a multi-pass transform; (0, 0) for lengths one kernel handles."""
    import ctypes

    lo, best = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)
    _check(_lib.load().tfft_workspace_bytes(dtype_code(dtype), int(N), int(batch), lo, best))
    return lo.value, best.value


def _points(t, N):
    """Number of complex points a tensor holds, and its TFFT dtype code."""
    if not t.is_cuda:
        raise ThunderFFTError(3, "ThunderFFT_global needs device tensors")
    if not t.is_contiguous():
        raise ThunderFFTError(3, "ThunderFFT_global needs contiguous [batch][N] interleaved data")
    code = _DTYPE_CODE.get(t.dtype)
    if code is None:
        raise ThunderFFTError(1, f"unsupported element type {t.dtype}")
    pts = t.numel() if t.dtype == torch.complex64 else t.numel() // 2
    return pts, code


def ThunderFFT_global(d_input, d_output, batch, N, forward=True, stream=None, options=None):
    """Launch `batch` transforms of length N from d_input to d_output (may be the same tensor).

    Tensors are complex64 [batch, N], or float32 / float16 / bfloat16 [batch, N, 2] with
    interleaved (re, im). Asynchronous on `stream` (default: torch's current stream).
    `options`: per-call launch options (see exec_options); every choice gives the same results.
    """
    pts_in, code = _points(d_input, N)
    pts_out, code_out = _points(d_output, N)
    if code != code_out:
        raise ThunderFFTError(3, "input and output element types differ")
    if pts_in < batch * N or pts_out < batch * N:
        raise ThunderFFTError(3, f"buffers hold {pts_in}/{pts_out} points, need {batch * N}")
    if stream is None:
        stream = torch.cuda.current_stream(d_input.device)
    with torch.cuda.device(d_input.device):
        if options:
            _check(_lib.load().tfft_exec_c2c_opt(code, int(N), int(bool(forward)), d_input.data_ptr(),
                                                 d_output.data_ptr(), int(batch), stream.cuda_stream,
                                                 exec_options(options)))
        else:
            _check(_lib.load().tfft_exec_c2c(code, int(N), int(bool(forward)), d_input.data_ptr(),
                                             d_output.data_ptr(), int(batch), stream.cuda_stream))
    return d_output


def bench_access_pattern(d_input, d_output, batch, N, pattern, forward=True, stream=None, options=None):
    """MEASUREMENT ONLY (the output is meaningless): the default kernel of (dtype, N) with its arithmetic
    (pattern 1) or its LDS exchanges (pattern 2) compiled out. See tfft_bench_access_pattern."""
    _, code = _points(d_input, N)
    if stream is None:
        stream = torch.cuda.current_stream(d_input.device)
    with torch.cuda.device(d_input.device):
        _check(_lib.load().tfft_bench_access_pattern(code, int(N), int(bool(forward)), int(pattern),
                                                     d_input.data_ptr(), d_output.data_ptr(), int(batch),
                                                     stream.cuda_stream, exec_options(options)))
    return d_output


def fft(x, forward=True, out=None):
    """Convenience: transform the last dimension of x (complex64 [.., N] or real [.., N, 2])."""
    N = x.shape[-1] if x.dtype == torch.complex64 else x.shape[-2]
    pts, code = _points(x, N)
    batch = pts // N
    ThunderFFTInitialize(code, N)
    if out is None:
        out = torch.empty_like(x)
    return ThunderFFT_global(x, out, batch, N, forward)
