"""The benchmark binaries keep the reference's command lines and the output lines its drivers parse.

thff_conv1d_fullfft4096: 11 positional arguments (thunderfft/sample/conv1d/conv1d_fullfft_l4096_thunder.cu:305-327)
and the line matched by Figure5_1dFFT/benchmark_figure5_1dfft.py:27-31; thff_conv1d: six arguments and the report
of sample/conv1d/main.cu:241-250, 303-304, 317-319; tfft_bench: the csv / ndjson sinks of sample/bench/util/stat.cu:53-77.
"""
import json
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# the regular expression of the reference's driver, verbatim in meaning: one float after "time:"
OURS_LINE = re.compile(
    r"ThunderFFT full FFT-Conv \(N=4096, include_filter_fft=[01], filter_fft_mode=[01]\) time:\s*([0-9.]+)\s*ms")


def run(*args):
    exe = os.path.join(ROOT, "bench", args[0])
    assert os.path.exists(exe), f"build {exe} with `make bench`"
    return subprocess.run([exe, *map(str, args[1:])], capture_output=True, text=True, timeout=120, check=True).stdout


@pytest.mark.parametrize("filter_fft_mode,real_input,real_filter", [(1, 0, 0), (0, 0, 0), (1, 1, 0), (1, 1, 1)])
def test_fullfft4096_cli_and_validation(filter_fft_mode, real_input, real_filter):
    out = run("thff_conv1d_fullfft4096", 9, 0, 3, 5, 1, 2, 1, 1, filter_fft_mode, real_input, real_filter)
    m = OURS_LINE.search(out)
    assert m and float(m.group(1)) > 0, out
    conv = re.search(r"Validation vs direct circular conv \(raw/N\): L2_rel=([0-9.eE+-]+) max_abs=([0-9.eE+-]+)", out)
    corr = re.search(r"Validation vs direct circular corr \(raw/N\): L2_rel=([0-9.eE+-]+)", out)
    assert conv and corr, out
    assert float(conv.group(1)) < 2e-3, out   # fp16 storage; the reference's own fp16 pipeline is not tighter
    assert float(corr.group(1)) > 0.05, out   # the result is the convolution, not the correlation


def test_fullfft4096_rejects_bad_tap_count():
    exe = os.path.join(ROOT, "bench", "thff_conv1d_fullfft4096")
    r = subprocess.run([exe, "5000"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "K must be in [1, 4096]" in r.stderr


@pytest.mark.parametrize("input_len,filter_len", [(65536, 9), (1000, 64), (64, 1)])
def test_tiled_conv_cli_and_validation(input_len, filter_len):
    out = run("thff_conv1d", input_len, filter_len, 0, 1, 1, 3)
    assert f"output_len : {input_len - filter_len + 1}" in out
    assert f"valid_tile : {64 - filter_len + 1}" in out
    assert re.search(r"Kernel pipeline time: [0-9.]+ ms", out), out
    m = re.search(r"Validation vs direct conv: L2_rel=([0-9.eE+-]+) Linf_rel=([0-9.eE+-]+)", out)
    assert m and float(m.group(1)) < 2e-3, out


def test_fft_sample_report():
    # the reference's minimal sample (sample/fft1d/main.cu:74-163): tones on bins 1 + b and 7
    out = run("thff_fft")
    m = re.search(r"Compare vs hipFFT\s*: L2_rel=([0-9.eE+-]+)\s+Linf_rel=([0-9.eE+-]+)", out)
    assert m and float(m.group(1)) < 1e-5 and float(m.group(2)) < 1e-5, out
    assert re.search(r"ThunderFFT avg time : [0-9.]+ ms", out) and "Checksum(ThunderFFT):" in out
    # batch 0: amplitude 0.5 on bin 1 and 0.1 on bin 7, times N = 64
    k1 = re.search(r"k=\s*1\s+TF=\(([-0-9.]+), ([-0-9.]+)\)", out)
    k7 = re.search(r"k=\s*7\s+TF=\(([-0-9.]+), ([-0-9.]+)\)", out)
    assert abs(float(k1.group(1)) - 32.0) < 1e-3 and abs(float(k7.group(1)) - 6.4) < 1e-3, out


def test_tfft_bench_sinks():
    out = run("tfft_bench", "--n", "256,4096", "--points", 1 << 22, "--runs", 2, "--ndjson", "--inside-repeats", 4)
    rows = [json.loads(line) for line in out.splitlines() if line.startswith("{")]
    ours = [r for r in rows if r["type"].startswith("thunderfft(")]
    assert [r["N"] for r in ours] == [256, 4096]
    for r in ours:
        assert r["max_abs_err"] < 1e-3 and r["e2e_ms"] > 0 and r["compute_ms"] > 0 and r["gflops"] > 0
    # the shared-memory variant of the reference's bench (baseline/thunderfft_bench.h:139-261): staged device calls
    staged = [r for r in rows if r["type"].startswith("thunderfft_s(")]
    assert [r["N"] for r in staged] == [256, 4096]
    for r in staged:
        assert r["e2e_ms"] > 0 and r["compute_ms"] > 0
    out = run("tfft_bench", "--n", "1024", "--points", 1 << 20, "--runs", 2, "--csv")
    assert out.splitlines()[0] == "type,N,radix,B,max_abs_err,compute_ms,e2e_ms,gflops,gbps,hbm_frac"


@pytest.mark.parametrize("extra", [[], ["--dtype", "fp16"]])
def test_bench_py_prints_one_json_line_with_the_contract_keys(extra):
    """bench.py --gpus 1 --steps K --warmup W: one JSON line on stdout with the driver's keys, the roofline object
    (achieved / peak = frac, measured on the launch stream) and, at one GPU, the CPU baseline object. A small batch
    keeps it short; the keys and their consistency are what is checked, not the rate."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--batch", "2048"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


def test_cmake_consumer_program():
    """examples/cmake_consumer/main.cpp, the program an outside CMake project builds on the `thunderfft` target
    (tests/test_cmake_build.py builds it that way on the CPU): Initialize, ThunderFFT_global, Finalize on tones."""
    exe = os.path.join(ROOT, "examples", "consumer")
    assert os.path.exists(exe), "build it with `make examples`"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"consumer: N=1024 batch=64 max_abs_err=([0-9.eE+-]+)", r.stdout)
    assert m and float(m.group(1)) < 1e-2, r.stdout
