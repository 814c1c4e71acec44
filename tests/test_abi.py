"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol the
header declares, and reports the plan table. No compute call is made (no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "thunderfft_c.h")


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tfft_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_match_binding():
    from fft_amd import _lib

    assert _declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_loads_and_exports_every_symbol():
    from fft_amd import _lib

    lib = _lib.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), name


def test_supported_table():
    import fft_amd

    for code in (fft_amd.TFFT_F32, fft_amd.TFFT_F16, fft_amd.TFFT_BF16):
        for n in fft_amd.SUPPORTED_N:
            assert fft_amd.is_supported(code, n, True)
            assert fft_amd.is_supported(code, n, False)  # the reference lacks inverse 512 / 2048
        for n in (0, 1, 32, 96, 1000, 3 << 20, 1 << 31):
            assert not fft_amd.is_supported(code, n if n < (1 << 31) else (1 << 31) - 1, True)
        # beyond one workgroup's LDS: multi-pass plans, fp32 only
        assert fft_amd.is_supported(code, 1 << 15, True) == (code == fft_amd.TFFT_F32)
        assert fft_amd.is_supported(code, 1 << 30, False) == (code == fft_amd.TFFT_F32)
    assert not fft_amd.is_supported(7, 1024, True)


def test_big_plan_factors():
    import fft_amd

    for lg in range(15, 31):
        info = fft_amd.plan_info("fp32", 1 << lg)
        prod = 1
        for r in info["radix"]:
            assert 64 <= r <= 1024
            prod *= r
        assert prod == 1 << lg
        assert len(info["radix"]) == (2 if lg <= 20 else 3)


def test_plan_info_geometry():
    import fft_amd

    for n in fft_amd.SUPPORTED_N:
        info = fft_amd.plan_info("fp32", n)
        assert info["n"] == n
        prod = 1
        for r in info["radix"]:
            prod *= r
        assert prod == n
        assert info["threads_per_block"] % 64 == 0
        assert info["points_per_thread"] * info["threads_per_block"] == n * info["transforms_per_block"]
        assert info["lds_bytes"] <= 160 * 1024


def test_errors_are_reported():
    import fft_amd

    with pytest.raises(fft_amd.ThunderFFTError) as e:
        fft_amd.plan_info("fp32", 1000)
    assert e.value.code == 1 and "1000" in str(e.value)
    with pytest.raises(fft_amd.ThunderFFTError):
        fft_amd.set_option("no_such_option", 1)


def test_no_option_can_select_a_wrong_result_build():
    """The measurement builds (kernel ids 1 and 2: work compiled out, meaningless output) are not plans: the
    transform entry point refuses them before it touches a device, the process-wide switch has no such key, and
    the measurement entry point takes only those two patterns."""
    import ctypes

    import fft_amd
    from fft_amd import _lib

    lib = _lib.load()
    for key in ("mode", "xcd_run_log2", "grid_blocks_per_cu"):
        with pytest.raises(fft_amd.ThunderFFTError):
            fft_amd.set_option(key, 1)
    buf = ctypes.create_string_buffer(64)  # never dereferenced: the options are rejected first
    ptr = ctypes.addressof(buf) & ~15
    for plan in (1, 2, -1):
        opts = fft_amd.exec_options({"plan": plan})
        assert lib.tfft_exec_c2c_opt(0, 4096, 1, ptr, ptr, 1, None, opts) == 3
        assert lib.tfft_conv_exec_opt(1, 4096, 0, ptr, ptr, ptr, 1, 1, None, opts) == 3
    bad = fft_amd.exec_options({"plan": 3})
    bad.struct_size = 4
    assert lib.tfft_exec_c2c_opt(0, 4096, 1, ptr, ptr, 1, None, bad) == 3
    assert lib.tfft_bench_access_pattern(0, 4096, 1, 5, ptr, ptr, 1, None, None) == 3
    assert lib.tfft_bench_access_pattern(0, 4096, 1, 3, ptr, ptr, 1, None, None) == 1  # patterns 3, 4: 16-bit kernels only
    assert lib.tfft_bench_access_pattern(0, 4096, 1, 1, ptr, ptr, 1, None, fft_amd.exec_options({"plan": 3})) == 3
    with pytest.raises(fft_amd.ThunderFFTError):
        fft_amd.exec_options({"no_such_field": 1})


def test_workspace_bytes():
    import fft_amd

    assert fft_amd.workspace_bytes("fp32", 4096, 100) == (0, 0)
    assert fft_amd.workspace_bytes("fp32", 1 << 20, 1) == (8 << 20, 8 << 20)
    assert fft_amd.workspace_bytes("fp32", 1 << 20, 1000) == (8 << 20, 8 << 26)  # the library's own round: 2^26 points
    assert fft_amd.workspace_bytes("fp32", 1 << 28, 3) == (8 << 28, 8 << 28)
    with pytest.raises(fft_amd.ThunderFFTError):
        fft_amd.workspace_bytes("fp16", 1 << 20, 1)


def test_host_wrapper_rejects_cpu_tensors():
    import torch

    import fft_amd

    x = torch.zeros(2, 64, dtype=torch.complex64)
    with pytest.raises(fft_amd.ThunderFFTError):
        fft_amd.ThunderFFT_global(x, x, 2, 64)


def test_synth_is_deterministic_and_bounded():
    import numpy as np

    from fft_amd import synth

    a = synth.uniform_complex(4, 256, seed=0)
    b = synth.uniform_complex(4, 256, seed=0)
    c = synth.uniform_complex(4, 256, seed=1)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert np.all(np.abs(a.real) <= 1) and np.all(np.abs(a.imag) <= 1)
    assert abs(a.real.mean()) < 0.1 and 0.5 < a.real.std() < 0.65


def test_reference_helper_names_compile_and_compute(tmp_path):
    """CHECK_CUDA, pad_h, reverse_bit_groups and swap_thread_data, the global-namespace names the reference's samples
    use beside the device API (detail/util_cuda.h:10-18, detail/util_arith.h:16-30, 123-167): a host program that
    includes the device header builds with hipcc and the helpers compute what the reference's do."""
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = tmp_path / "helpers.hip"
    src.write_text(r'''
#include <thunderfft/thunderfft_device.hpp>
#include <cstdio>
int main() {
  static_assert(reverse_bit_groups<2, 6>(0b000110) == 0b100100, "base-4 digit reversal");
  static_assert(reverse_bit_groups<2, 7>(0b1000001) == 33, "leftover bit follows");
  static_assert(pad_h(100) == -1 && pad_h(4096) > 0, "pad_h");
  static_assert(8 * (4096 + pad_h(4096)) * thunderfft::batch_per_block<4096> >= thunderfft::workspace_bytes<4096>, "");
  static_assert(8 * (256 + pad_h(256)) * thunderfft::batch_per_block<256> >= thunderfft::workspace_bytes<256>, "");
  int v[32];
  for (int i = 0; i < 32; ++i) v[i] = i;
  swap_thread_data(v);
  CHECK_CUDA(hipSuccess);
  std::printf("%d %d %d %d %d %d\n", v[0], v[1], v[4], v[17], v[27], v[31]);
  return 0;
}
''')
    exe = tmp_path / "helpers"
    lib_dir = os.path.join(ROOT, "fft_amd", "lib")
    subprocess.run([hipcc, "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-I", os.path.join(ROOT, "include"),
                    "-o", str(exe), str(src), "-L", lib_dir, "-lthunderfft_hip", f"-Wl,-rpath,{lib_dir}"],
                   check=True, capture_output=True, timeout=600)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, timeout=60).stdout
    assert out.split() == ["0", "4", "1", "20", "30", "31"]
