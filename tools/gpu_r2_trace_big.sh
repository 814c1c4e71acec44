#!/bin/bash
# Per-sweep kernel times of the multi-pass transforms (rocprofv3 kernel trace), one length per run.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_big
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for n in ${LENGTHS:-1048576 16777216 67108864}; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/t$n -o b -- $R/bench/tfft_bench --n $n --points 67108864 --no-hipfft --runs 5 --inplace 0 > $OUT/t$n.log 2>&1 || { tail -5 $OUT/t$n.log; exit 1; }
  python3 - $OUT/t$n/b_kernel_trace.csv $n <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "fft_tile_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed launches are the last 5 transforms; print the sweeps of the last one
per = collections.OrderedDict()
for r in rows:
    key = (r["Kernel_Name"][:110], r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", ""))
    per.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("n =", sys.argv[2])
for (k, g), v in per.items():
    v = v[-5:]
    print(f"  {sum(v)/len(v):8.1f} us  grid={g}  {k}")
PY
done
