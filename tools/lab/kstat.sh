#!/bin/bash
# lab: average duration of the kernels matching $1 in a profiled bench run
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/kstat
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o b -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --particles 0 > $O/log 2>&1
f=$(find $O -name "b_kernel_stats.csv" | head -1)
grep -E "$1" $f | cut -d, -f1-4 | cut -c 1-150
tail -1 $O/log | cut -c 1-200
