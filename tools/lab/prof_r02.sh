#!/bin/bash
# The profile recipe behind profiles/r02_*: run on the GPU box from the repo root
#   (gpurun -- 'bash tools/lab/prof_r02.sh'); outputs land under gpurun_out/prof_r02/, and
#   tools/lab/prof_r02_summary.py turns them into the files committed under profiles/.
# 1. kernel trace + stats of the bench command (with the config D particle lines)
# 2./3. FETCH_SIZE / WRITE_SIZE of the relax kernels, in their own passes with --kernel-trace only
# 4. the bench line itself, un-profiled
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o b -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/trace.log 2>&1 && \
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 $R/tools/relax_only.py 8 > $O/pmc_fetch.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 $R/tools/relax_only.py 8 > $O/pmc_write.log 2>&1
cd $R && timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
