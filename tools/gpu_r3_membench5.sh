#!/bin/bash
# membench5 (strided tile copies with row segments below a cache line): timings, then FETCH_SIZE / WRITE_SIZE of
# the variants named in PMC_VARIANTS, each counter in a pass of its own.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3_membench5
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 120 $R/bench/membench5 --lg ${LG:-26} --runs 10 > $OUT/timing.txt 2>&1 || { tail -5 $OUT/timing.txt; exit 1; }
cat $OUT/timing.txt
for v in ${PMC_VARIANTS:-r2048_s128 r2048_s64_o0 r2048_s64_o1 r2048_s32_o0 r2048_s32_o1 r4096_s32_o0 r4096_s32_o1 r8192_s32_o1 r8192_s16_o1 r1024_s128}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 60 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${v}_$c -o m -- $R/bench/membench5 --lg ${LG:-26} --runs 2 --only $v > $OUT/pmc_${v}_$c.log 2>&1 || { tail -3 $OUT/pmc_${v}_$c.log; exit 1; }
  done
  python3 - "$OUT" "$v" "${LG:-26}" <<'PY'
import csv, glob, sys
out, v, lg = sys.argv[1], sys.argv[2], int(sys.argv[3])
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = [float(r["Counter_Value"]) for p in glob.glob(f"{out}/pmc_{v}_{c}/**/*counter_collection.csv", recursive=True)
            for r in csv.DictReader(open(p)) if "tile_copy" in r["Kernel_Name"] and r["Counter_Name"] == c]
    vals[c] = sum(rows) / len(rows) * 1024
alg = 8 << lg
print(f"{v:20s} FETCH_SIZE {vals['FETCH_SIZE']/alg:.3f} x (raw KiB counter; wide coalesced reads tally at half)  "
      f"WRITE_SIZE {vals['WRITE_SIZE']/alg:.3f} x of the {alg >> 20} MiB moved each way")
PY
done
