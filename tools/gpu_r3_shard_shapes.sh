#!/bin/bash
# Per-rank shard shapes of the strong-scaling sweep on one GPU: the 2^28-point batch of length N split over
# G = 1, 2, 4, 8 ranks is, for each rank, a launch of batch / G transforms. Timing those launches on one GPU gives
# the strong-scaling efficiency batch sharding can reach (no collective is on the data path): t(1) / (G t(G)).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3_shard_shapes.jsonl
: > $OUT
for n in ${LENGTHS:-16384 4096 256}; do
  for g in 1 2 4 8; do
    batch=$(( (1 << 28) / n / g ))
    timeout -k 10 120 python3 $R/bench.py --fft-len $n --batch $batch --steps 50 --warmup 5 --no-cpu-baseline >> $OUT || exit 1
    echo "n=$n G=$g batch=$batch done"
  done
done
python3 - "$OUT" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
base = {}
for r in rows:
    n = int(r["config"]["workload"].split("N=")[1].split()[0]); b = r["config"]["total_batch"]
    g = (1 << 28) // n // b
    if g == 1: base[n] = r["roofline"]["kernel_ms"]
    print(f"N={n:6d} G={g} batch/rank={b:8d} kernel_ms={r['roofline']['kernel_ms']:.4f} frac={r['roofline']['frac']:.4f} "
          f"implied strong-scaling efficiency={base[n] / (g * r['roofline']['kernel_ms']):.3f}")
PY
