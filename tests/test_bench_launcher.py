"""bench.py's multi-rank launcher on the CPU: `--gpus N` with no WORLD_SIZE starts N ranks of itself under
torch.distributed.run as a child process and relays rank 0's line. The step is a stub (`--stub-step`: gloo
ranks, a sleep instead of a kernel; the line says so in `data`), so what is checked is the launch plumbing:
rank count, scaling modes, sharding of the batch, one line on stdout."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run_bench(*args, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *map(str, args)], capture_output=True, text=True, timeout=300,
                          cwd=ROOT, env=e)


def one_line(out):
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks_weak():
    d = one_line(run_bench("--gpus", 2, "--steps", 3, "--warmup", 1, "--stub-step"))
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3 and d["warmup"] == 1
    assert d["data"].startswith("stub")
    # weak scaling: every rank holds the whole per-GPU workload (2^28 points at the default length)
    assert d["config"]["points_per_gpu"] == 1 << 28 and d["config"]["total_batch"] == 2 * 65536
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only


def test_gpus_2_strong_scaling_splits_the_batch():
    d = one_line(run_bench("--gpus", 2, "--steps", 2, "--warmup", 0, "--stub-step", "--scaling", "strong", "--n", 16384))
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["total_batch"] == 16384 and d["config"]["points_per_gpu"] == 1 << 27
    assert "N=16384" in d["metric"]


def test_single_rank_stub_and_mismatched_world():
    d = one_line(run_bench("--gpus", 1, "--steps", 2, "--warmup", 0, "--stub-step"))
    assert d["n_gpus"] == 1 and d["config"]["points_per_gpu"] == 1 << 28
    out = run_bench("--gpus", 2, "--stub-step", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr


def test_under_torchrun_directly():
    """The driver's own form for N > 1: torch.distributed.run starts the ranks, bench.py is one of them."""
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29731", BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--stub-step"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=e)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{"metric"')]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-3000:]
    assert json.loads(lines[0])["n_gpus"] == 2
