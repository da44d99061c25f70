cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof27 -o b -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --particles 0 > $R/gpurun_out/prof27.log 2>&1
head -8 $R/gpurun_out/prof27/b_kernel_stats.csv | cut -c1-120
