#!/bin/bash
# bench.py with one event pair around the timed region (default) against one pair per launch, plain and under
# rocprofv3 --kernel-trace (whose per-kernel durations are the kernels' own begin-to-end times).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3_event_modes
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
pick='import sys,json; r=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(r["config"]["workload"][:60], "frac", r["roofline"]["frac"], "kernel_ms", r["roofline"]["kernel_ms"], "ms_per_step", r["ms_per_step"])'
for args in "" "--fft-len 16384" "--dtype bf16" "--dtype fp16"; do
  for ev in "" "--event-per-step"; do
    echo "== bench.py $args $ev"
    timeout -k 10 120 python3 $R/bench.py $args $ev --steps 20 --warmup 3 --no-cpu-baseline | python3 -c "$pick" || exit 1
  done
done
for ev in "" "--event-per-step"; do
  tag=region; [ -n "$ev" ] && tag=perstep
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -o b -- python3 $R/bench.py $ev --steps 20 --warmup 3 --no-cpu-baseline > $OUT/trace_$tag.log 2>&1 || { tail -5 $OUT/trace_$tag.log; exit 1; }
  echo "== under rocprofv3, $tag"; grep '^{' $OUT/trace_$tag.log | python3 -c "$pick"
  python3 - $OUT/trace_$tag <<'PY'
import csv, glob, sys
rows = [r for p in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(p))]
fft = sorted((r for r in rows if "fft_lds_kernel" in r["Kernel_Name"] and "true, 0>" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
fft = fft[3:]  # warm-up launches
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in fft]
gap = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(fft, fft[1:])]
print(f"   {len(fft)} timed launches: kernel begin-to-end mean {sum(dur)/len(dur)/1e3:.1f} us, gap between launches mean {sum(gap)/len(gap)/1e3:.1f} us, "
      f"first-begin to last-end / launches {(int(fft[-1]['End_Timestamp']) - int(fft[0]['Start_Timestamp'])) / len(fft) / 1e3:.1f} us")
PY
done
