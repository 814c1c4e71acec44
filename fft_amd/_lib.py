"""ctypes binding of libthunderfft_hip.so (C ABI declared in include/thunderfft_c.h)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# THUNDERFFT_HIP_LIB: another build of the same library (A/B timing of two builds, an installed copy); it must export
# the same C ABI. There is still no fallback: the named file is the one that is loaded, or the import fails.
LIB_PATH = os.environ.get("THUNDERFFT_HIP_LIB") or os.path.join(_HERE, "lib", "libthunderfft_hip.so")

# Every symbol include/thunderfft_c.h declares; tests check the list against the header.
SYMBOLS = (
    "tfft_initialize",
    "tfft_finalize",
    "tfft_exec_c2c",
    "tfft_exec_c2c_opt",
    "tfft_exec_c2c_static",
    "tfft_exec_options_init",
    "tfft_workspace_bytes",
    "tfft_bench_access_pattern",
    "tfft_bench_onchip_repeats",
    "tfft_is_supported",
    "tfft_get_plan_info",
    "tfft_set_option",
    "tfft_fourstep_twiddle",
    "tfft_dist_permute",
    "tfft_dist_radix",
    "tfft_conv_exec",
    "tfft_conv_exec_opt",
    "tfft_conv_filter_layout",
    "tfft_conv_filter_layout_opt",
    "tfft_conv_filter_spectra",
    "tfft_conv_forget_layout",
    "tfft_conv_tiles_exec",
    "tfft_get_twiddle_table",
    "tfft_last_error",
)


class PlanInfo(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_int),
        ("threads_per_block", ctypes.c_int),
        ("transforms_per_block", ctypes.c_int),
        ("points_per_thread", ctypes.c_int),
        ("radix", ctypes.c_int * 3),
        ("lds_bytes", ctypes.c_int),
        ("blocks_per_cu", ctypes.c_int),
    ]


class ExecOptions(ctypes.Structure):
    """tfft_exec_options of include/thunderfft_c.h."""
    _fields_ = [
        ("struct_size", ctypes.c_int),
        ("plan", ctypes.c_int),
        ("xcd_run_log2", ctypes.c_int),
        ("grid_blocks_per_cu", ctypes.c_int),
        ("grid_blocks", ctypes.c_longlong),
        ("extra_lds", ctypes.c_int),
        ("reserved", ctypes.c_int),
        ("workspace", ctypes.c_void_p),
        ("workspace_bytes", ctypes.c_ulonglong),
    ]


class DistPermuteArgs(ctypes.Structure):
    """tfft_dist_permute_args of include/thunderfft_c.h."""
    _fields_ = [(k, ctypes.c_longlong) for k in ("nb1", "nb2", "ni", "nj", "in_b1", "in_b2", "in_i",
                                                  "out_b1", "out_b2", "out_j")] + [
        ("transpose", ctypes.c_int), ("twiddle", ctypes.c_int), ("tw_n", ctypes.c_longlong),
        ("tw_shift", ctypes.c_int), ("reserved", ctypes.c_int)] + [
        (k, ctypes.c_longlong) for k in ("tw_p0", "tw_pi", "tw_q0", "tw_qb1", "tw_qb2", "tw_qj")]


class DistRadixArgs(ctypes.Structure):
    """tfft_dist_radix_args of include/thunderfft_c.h."""
    _fields_ = [("parts", ctypes.c_int), ("reserved", ctypes.c_int)] + [
        (k, ctypes.c_longlong) for k in ("rows", "row_len", "in_chunk", "in_pitch", "out_chunk", "out_pitch")]


_lib = None


def load():
    """Load the HIP library. There is no fallback: a missing or unloadable library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make lib` (or __graft_entry__.build()); "
            "fft_amd has no CPU or eager fallback")
    # torch bundles its own libamdhip64.so.7; importing it first makes this library resolve
    # the same HIP runtime, so device pointers and streams are shared with torch.
    import torch  # noqa: F401

    lib = ctypes.CDLL(LIB_PATH)
    i, ll, vp, cp = ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_char_p
    lib.tfft_initialize.argtypes = [i, i]
    lib.tfft_finalize.argtypes = [i]
    lib.tfft_exec_c2c.argtypes = [i, i, i, vp, vp, ll, vp]
    op = ctypes.POINTER(ExecOptions)
    ullp = ctypes.POINTER(ctypes.c_ulonglong)
    lib.tfft_exec_c2c_opt.argtypes = [i, i, i, vp, vp, ll, vp, op]
    lib.tfft_exec_c2c_static.argtypes = [i, i, i, vp, vp, ll, vp]
    lib.tfft_exec_options_init.argtypes = [op]
    lib.tfft_exec_options_init.restype = None
    lib.tfft_workspace_bytes.argtypes = [i, i, ll, ullp, ullp]
    lib.tfft_bench_access_pattern.argtypes = [i, i, i, i, vp, vp, ll, vp, op]
    lib.tfft_bench_onchip_repeats.argtypes = [i, i, i, vp, vp, ll, vp]
    lib.tfft_conv_exec_opt.argtypes = [i, i, i, vp, vp, vp, i, i, vp, op]
    lib.tfft_conv_filter_layout_opt.argtypes = [i, i, vp, vp, i, vp, op]
    lib.tfft_is_supported.argtypes = [i, i, i]
    lib.tfft_get_plan_info.argtypes = [i, i, ctypes.POINTER(PlanInfo)]
    lib.tfft_set_option.argtypes = [cp, ll]
    lib.tfft_fourstep_twiddle.argtypes = [i, vp, ll, ll, ll, ll, ll, i, vp]
    lib.tfft_dist_permute.argtypes = [vp, vp, ctypes.POINTER(DistPermuteArgs), vp, vp, i, vp]
    lib.tfft_dist_radix.argtypes = [vp, vp, ctypes.POINTER(DistRadixArgs), vp, i, vp]
    lib.tfft_conv_exec.argtypes = [i, i, i, vp, vp, vp, i, i, vp]
    lib.tfft_conv_filter_layout.argtypes = [i, i, vp, vp, i, vp]
    lib.tfft_conv_filter_spectra.argtypes = [i, i, vp, vp, i, vp]
    lib.tfft_conv_forget_layout.argtypes = [vp, ctypes.c_ulonglong]
    lib.tfft_conv_tiles_exec.argtypes = [i, i, vp, vp, vp, ll, ll, i, vp]
    lib.tfft_get_twiddle_table.argtypes = [i, i, ctypes.POINTER(vp)]
    lib.tfft_last_error.argtypes = []
    lib.tfft_last_error.restype = cp
    for name in SYMBOLS:
        if name not in ("tfft_last_error", "tfft_exec_options_init"):
            getattr(lib, name).restype = i
    _lib = lib
    return lib
