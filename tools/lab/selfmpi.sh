#!/bin/bash
# lab: the code path of a box in an N > 1 run (all sides GfsBoundaryMpi, RCCL to self) on one GPU
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/selfmpi
mkdir -p $O
cd $R && timeout -k 10 200 python bench.py --self-mpi --steps 5 --warmup 2 --no-cpu-baseline --particles 0 > $O/bench.json 2> $O/bench.err
cat $O/bench.json | cut -c 1-400
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o b -- python3 $R/bench.py --self-mpi --steps 5 --warmup 1 --no-cpu-baseline --particles 0 > $O/trace.log 2>&1
python3 $R/tools/lab/step_breakdown.py $O/trace advect > $O/breakdown.txt 2>&1
cat $O/breakdown.txt
