#!/bin/bash
# Per-kernel times of the distributed transform's local passes (eight virtual ranks on one GPU).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_dist
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o b -- python3 $R/tools/bench_distributed.py --log2n ${LOG2N:-28} --virtual-ranks 8 --steps 3 > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - $OUT/t/b_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f'{float(r["AverageNs"])/1e3:9.1f} us x{r["Calls"]:>5}  {float(r["Percentage"]):5.1f}%  {r["Name"][:110]}')
PY
