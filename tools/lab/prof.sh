cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof44 -o b -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --particles 0 > $R/gpurun_out/prof44.log 2>&1
tail -c 300 $R/gpurun_out/prof44.log | head -c 10
