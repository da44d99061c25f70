cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof42 -o b -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --particles 0 > $R/gpurun_out/prof42.log 2>&1 && \
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc42f -o f -- python3 $R/tools/relax_only.py 8 > $R/gpurun_out/pmc42f.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc42w -o w -- python3 $R/tools/relax_only.py 8 > $R/gpurun_out/pmc42w.log 2>&1
cd $R && timeout -k 10 300 python bench.py > gpurun_out/bench42.json 2> gpurun_out/bench42.err
tail -c 600 gpurun_out/bench42.json
