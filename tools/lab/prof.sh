cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof35 -o b -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --particles 0 > $R/gpurun_out/prof35.log 2>&1
cd $R && timeout -k 10 300 python bench.py > gpurun_out/bench35.json 2> gpurun_out/bench35.err
cat gpurun_out/bench35.json
