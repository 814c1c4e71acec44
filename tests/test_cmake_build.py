"""The CMake build a consumer of the reference would use: add_subdirectory + the `thunderfft` target, and the
reference's sample target names at the reference's output paths (thunderfft/CMakeLists.txt:23-30,
sample/*/CMakeLists.txt, Figure5_1dFFT/run_figure5.sh:15-16, 36). Runs without a GPU: configures with the in-tree
library as THUNDERFFT_PREBUILT_LIB, so only host-side sample code is compiled (seconds)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "fft_amd", "lib", "libthunderfft_hip.so")

pytestmark = pytest.mark.skipif(shutil.which("cmake") is None or shutil.which("ninja") is None,
                                reason="cmake / ninja not installed")


def run(cmd, **kw):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, **kw)
    assert p.returncode == 0, " ".join(cmd) + "\n" + p.stdout[-3000:]
    return p.stdout


def test_figure5_target_and_consumer_build(tmp_path):
    build = str(tmp_path / "ref_style")
    run(["cmake", "-S", ROOT, "-B", build, "-G", "Ninja", f"-DTHUNDERFFT_PREBUILT_LIB={LIB}"])
    # the Figure-5 script's two steps: configure, then build this one target and run it from this path
    run(["cmake", "--build", build, "--target", "thff_conv1d_fullfft4096", "-j", "8"])
    exe = os.path.join(build, "sample", "conv1d", "thff_conv1d_fullfft4096")
    assert os.access(exe, os.X_OK)
    # the reference's other sample targets exist under their names
    targets = run(["cmake", "--build", build, "--target", "help"])
    for name in ("benchmark", "thff_fft", "thff_conv1d", "thff_conv1d_fullfft4096"):
        assert name in targets, name
    # the binary links the library through the `thunderfft` target and finds it at run time
    assert "libthunderfft_hip.so" in run(["ldd", exe])

    # an outside project: add_subdirectory(<this repo>) + target_link_libraries(app thunderfft)
    cons = str(tmp_path / "consumer")
    run(["cmake", "-S", os.path.join(ROOT, "examples", "cmake_consumer"), "-B", cons, "-G", "Ninja",
         f"-DTHUNDERFFT_PREBUILT_LIB={LIB}"])
    run(["cmake", "--build", cons])
    assert os.access(os.path.join(cons, "consumer"), os.X_OK)
