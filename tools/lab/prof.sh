#!/bin/bash
# The profile recipe behind profiles/r01_*: run on the GPU box from the repo root
#   (gpurun -- 'bash tools/lab/prof.sh'); outputs land under gpurun_out/.
# 1. kernel trace + stats of the bench command   2./3. FETCH_SIZE / WRITE_SIZE of the relax kernels
# (counters in their own passes, with --kernel-trace only)   4. the bench line itself
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o b -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --particles 0 > $R/gpurun_out/prof.log 2>&1 && \
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python3 $R/tools/relax_only.py 8 > $R/gpurun_out/pmc_fetch.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o w -- python3 $R/tools/relax_only.py 8 > $R/gpurun_out/pmc_write.log 2>&1
cd $R && timeout -k 10 300 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -c 400 gpurun_out/bench.json
