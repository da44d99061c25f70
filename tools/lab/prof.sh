cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof25 -o b -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --particles 0 > $R/gpurun_out/prof25.log 2>&1 && \
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc25f -o f -- python3 $R/tools/relax_only.py 8 > $R/gpurun_out/pmc25f.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc25w -o w -- python3 $R/tools/relax_only.py 8 > $R/gpurun_out/pmc25w.log 2>&1
ls $R/gpurun_out/prof25 $R/gpurun_out/pmc25f $R/gpurun_out/pmc25w
