#!/bin/bash
# 16-bit storage sweep at 2^28 points on a settled device: the matrix-core kernels take tens of milliseconds of
# sustained work to reach their rate (3 warm-up + 20 timed launches read 67-70 % where 50 + 100 read 78-80 %), so
# every line here is bench.py with 50 warm-up and 100 timed launches. fp16 / bf16, x -> y and in place.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3_sweep16.jsonl
: > $OUT
for dt in ${DTYPES:-fp16 bf16}; do for place in "" "--in-place"; do for n in ${LENGTHS:-256 512 1024 2048 4096 8192 16384}; do
  timeout -k 10 120 python3 $R/bench.py --dtype $dt --fft-len $n $place --warmup ${WARMUP:-50} --steps ${STEPS:-100} --no-cpu-baseline >> $OUT || exit 1
done; done; done
python3 - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        r = json.loads(l)
        print(f"{r['config']['workload'][:78]:78s} {r['roofline']['kernel_ms']:.4f} ms  {100 * r['roofline']['frac']:.1f} % of 8 TB/s  {r['value'] / 1e3:.1f} TFLOP/s")
PY
