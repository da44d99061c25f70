#!/bin/bash
# lab: average duration of the kernels matching $1 in a profiled bench run
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/kstat${2:+_$2}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o b -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --particles 0 > $O/log 2>&1
f=$(find $O -name "b_kernel_stats.csv" | head -1)
python3 - "$f" "$1" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print("%-60s calls %4s avg %9.1f us  min %9.1f  max %9.1f" % (r["Name"][:60], r["Calls"],
              float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
grep '^{"metric"' $O/log | cut -c 1-200
