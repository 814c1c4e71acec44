"""Host-side mirror of the reference's fused FFT-convolution sample, above the C ABI.

Reference: thunderfft/sample/conv1d/conv1d_fullfft_l4096_thunder.cu. Its main() prepares the
filter spectra (filter_fft_once, :412-426: forward FFT of the zero-padded, 1/N-scaled taps, then
permute_filter_layout4096) and launches one of three fused kernels per run (run_once, :433-449).
The same two steps are `filter_spectra` and `fft_conv` here; tensors stand in for device pointers.
"""
import torch

from . import _lib
from .api import ThunderFFTError, ThunderFFTInitialize, ThunderFFT_global, _check, dtype_code, exec_options

CONV_COMPLEX, CONV_REAL_IN, CONV_REAL_IN_OUT = 0, 1, 2
SUPPORTED_CONV_N = (4096,)


def _stream(t, stream):
    return (stream or torch.cuda.current_stream(t.device)).cuda_stream


def filter_layout(h_freq, N, stream=None, options=None):
    """[D, N, 2] natural-order filter spectra -> the register order the fused kernel reads (the kernel that
    `options` selects: pass the same options to fft_conv)."""
    if not (h_freq.is_cuda and h_freq.is_contiguous()) or h_freq.shape[-2:] != (N, 2):
        raise ThunderFFTError(3, "filter_layout needs a contiguous device tensor [D, N, 2]")
    D = h_freq.numel() // (2 * N)
    out = torch.empty_like(h_freq)
    with torch.cuda.device(h_freq.device):
        _check(_lib.load().tfft_conv_filter_layout_opt(dtype_code(h_freq.dtype), int(N), h_freq.data_ptr(),
                                                       out.data_ptr(), int(D), _stream(h_freq, stream),
                                                       exec_options(options)))
    return out


def forget_layout(buf):
    """Drops what the library recorded about the register order `buf` holds (see tfft_conv_forget_layout): call it
    after refilling a permuted-filter buffer by other means than filter_layout."""
    with torch.cuda.device(buf.device):
        _check(_lib.load().tfft_conv_forget_layout(buf.data_ptr(), buf.numel() * buf.element_size()))


def filter_spectra(h_time, N, stream=None, options=None):
    """Filter taps [D, K, 2] (already scaled by 1/N if the caller wants a normalised result, as the
    reference does at :352-361) -> register-order spectra [D, N, 2] of the same element type."""
    D, K = h_time.shape[0], h_time.shape[1]
    if K > N:
        raise ThunderFFTError(3, f"filter has {K} taps, transform length is {N}")
    padded = torch.zeros(D, N, 2, dtype=h_time.dtype, device=h_time.device)
    padded[:, :K] = h_time
    ThunderFFTInitialize(h_time.dtype, N)
    if not options and h_time.dtype == torch.float16:
        # the default (matrix-core) kernel: transform and re-ordering in one kernel (tfft_conv_filter_spectra)
        return filter_spectra_fused(padded, N, stream)
    spec = torch.empty_like(padded)
    ThunderFFT_global(padded, spec, D, N, True, stream)
    return filter_layout(spec, N, stream, options)


def filter_spectra_fused(padded, N, stream=None):
    """Zero-padded taps [D, N, 2] -> register-order spectra in one kernel (tfft_conv_filter_spectra): bit for bit
    ThunderFFT_global followed by filter_layout. fp16, N = 4096 (ThunderFFTError UNSUPPORTED otherwise)."""
    if not (padded.is_cuda and padded.is_contiguous()) or padded.shape[-2:] != (N, 2):
        raise ThunderFFTError(3, "filter_spectra_fused needs a contiguous device tensor [D, N, 2]")
    D = padded.numel() // (2 * N)
    out = torch.empty_like(padded)
    with torch.cuda.device(padded.device):
        _check(_lib.load().tfft_conv_filter_spectra(dtype_code(padded.dtype), int(N), padded.data_ptr(), out.data_ptr(),
                                                    int(D), _stream(padded, stream)))
    return out


def fft_conv(x, h_perm, N, kind=CONV_COMPLEX, out=None, stream=None, options=None):
    """y[b, d] = IDFT_N(DFT_N(x[b, d]) * H[d]), unnormalised, in one kernel.

    x: [B, D, N, 2] (complex kinds) or [B, D, N] (real-input kinds); h_perm from filter_spectra /
    filter_layout, [D, N, 2]; result [B, D, N, 2], or [B, D, N] for CONV_REAL_IN_OUT.
    """
    real_in = kind in (CONV_REAL_IN, CONV_REAL_IN_OUT)
    want = (N,) if real_in else (N, 2)
    if not (x.is_cuda and x.is_contiguous()) or x.dim() != 2 + len(want) or tuple(x.shape[2:]) != want:
        raise ThunderFFTError(3, f"fft_conv needs a contiguous device tensor [B, D, {', '.join(map(str, want))}]")
    B, D = x.shape[0], x.shape[1]
    if h_perm.shape[0] != D or h_perm.dtype != x.dtype or not h_perm.is_contiguous():
        raise ThunderFFTError(3, "filter spectra must be contiguous [D, N, 2] of the signal's element type")
    if out is None:
        shape = (B, D, N) if kind == CONV_REAL_IN_OUT else (B, D, N, 2)
        out = torch.empty(shape, dtype=x.dtype, device=x.device)
    ThunderFFTInitialize(x.dtype, N)
    with torch.cuda.device(x.device):
        _check(_lib.load().tfft_conv_exec_opt(dtype_code(x.dtype), int(N), int(kind), x.data_ptr(), h_perm.data_ptr(),
                                              out.data_ptr(), int(B), int(D), _stream(x, stream),
                                              exec_options(options)))
    return out


def tiled_filter_spectrum(taps, N=64):
    """Filter taps [K, 2] -> the spectrum the tiled kernel multiplies by: taps stored at (N - i) % N and
    scaled by 1/N (the reference's make_padded_filter_fft64, sample/conv1d/main.cu:145-154), transformed
    in fp32 and rounded to the taps' element type (the reference: cuFFT fp32, then f32_to_half2, :280-287)."""
    K = taps.shape[0]
    if K > N:
        raise ThunderFFTError(3, f"filter has {K} taps, tile transform length is {N}")
    padded = torch.zeros(1, N, 2, dtype=torch.float32, device=taps.device)
    idx = (N - torch.arange(K, device=taps.device)) % N
    padded[0, idx] = taps.float() / N
    ThunderFFTInitialize(torch.float32, N)
    spec = torch.empty_like(padded)
    ThunderFFT_global(padded, spec, 1, N, True)
    return spec[0].to(taps.dtype).contiguous()


def conv1d_tiled(x, hf, filter_len, N=64, out=None, stream=None):
    """Valid cross-correlation out[i] = sum_k x[i + k] * filter[k] by overlap-save tiles of length N.

    x: [L, 2] interleaved complex (float32 or float16) on the device; hf from tiled_filter_spectrum;
    result [L - filter_len + 1, 2]. Mirrors the launch in sample/conv1d/main.cu:235-299."""
    if not (x.is_cuda and x.is_contiguous()) or x.dim() != 2 or x.shape[1] != 2:
        raise ThunderFFTError(3, "conv1d_tiled needs a contiguous device tensor [L, 2]")
    L = x.shape[0]
    if filter_len < 1 or filter_len > N or L < filter_len:
        raise ThunderFFTError(3, f"need 1 <= filter_len <= {N} <= input length")
    if hf.dtype != x.dtype or hf.numel() != 2 * N or not hf.is_contiguous():
        raise ThunderFFTError(3, "filter spectrum must be contiguous [N, 2] of the signal's element type")
    out_len = L - filter_len + 1
    if out is None:
        out = torch.empty(out_len, 2, dtype=x.dtype, device=x.device)
    ThunderFFTInitialize(x.dtype, N)
    with torch.cuda.device(x.device):
        _check(_lib.load().tfft_conv_tiles_exec(dtype_code(x.dtype), int(N), x.data_ptr(), hf.data_ptr(), out.data_ptr(),
                                                int(L), int(out_len), int(N - filter_len + 1), _stream(x, stream)))
    return out
