#!/bin/bash
# bench.py lines of every configuration, as the driver runs it (3 warm-up + 20 timed launches) and on a settled
# device (50 + 100): gpurun_out/r3_bench_py_lines.jsonl, with a table.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3_bench_py_lines.jsonl
: > $OUT
run() { timeout -k 10 200 python3 $R/bench.py "$@" >> $OUT || exit 1; }
run                                   # the headline, with the CPU baseline
for ws in "3 20" "50 100"; do
  set -- $ws
  for n in 256 512 1024 2048 4096 8192 16384; do run --fft-len $n --warmup $1 --steps $2 --no-cpu-baseline; done
  run --dtype fp16 --warmup $1 --steps $2 --no-cpu-baseline
  run --dtype bf16 --warmup $1 --steps $2 --no-cpu-baseline
  run --fft-len 67108864 --batch 1 --warmup $1 --steps $2 --no-cpu-baseline
done
run --fft-len 1024 --batch 1024 --no-cpu-baseline    # the reference's small configuration
python3 - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        r = json.loads(l)
        ro = r["roofline"]
        extra = f" per-sweep {100 * ro['frac_sweep_model']:.1f} %" if "frac_sweep_model" in ro else ""
        cpu = f"  cpu {r['cpu_baseline']['value']} GFLOP/s on {r['cpu_baseline']['cores']} cores" if "cpu_baseline" in r else ""
        print(f"{r['config']['workload'][:66]:66s} w{r['warmup']:3d}+{r['steps']:3d}  {ro['kernel_ms']:.4f} ms  {100 * ro['frac']:.1f} %{extra}  "
              f"{r['value'] / 1e3:.2f} TFLOP/s  traffic {ro['traffic']}{cpu}")
PY
