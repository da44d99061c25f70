#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for T in 1024 512 256; do
  export GFSHIP_COARSE_THREADS=$T
  O=$R/gpurun_out/ct_$T
  rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o b -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --particles 0 > $O/log 2>&1
  f=$(find $O -name "b_kernel_stats.csv" | head -1)
  echo "threads $T: $(grep coarse_cycle_kernel $f | cut -d, -f1-5)"
done
