#!/bin/bash
# Kernel durations of the fused-convolution benchmark at the Figure-5 shape (rocprofv3 kernel trace).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_conv
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o b -- $R/bench/thff_conv1d_fullfft4096 ${CONV_ARGS:-33 0 64 768 5 10 1 0 1 1 1} > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - $OUT/t/b_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f'{float(r["AverageNs"])/1e3:9.1f} us x{r["Calls"]:>5}  {float(r["Percentage"]):5.1f}%  {r["Name"][:120]}')
PY
grep "time:" $OUT/log.txt
