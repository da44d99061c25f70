#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; timeout -k 10 600 python -m pytest tests/test_gpu_tree.py -x -q -m gpu > gpurun_out/tree_tests.log 2>&1; tail -2 gpurun_out/tree_tests.log
export TREE_BENCH_ARGS="2 7 2 3" TREE_BENCH_LOOPS=20
for W in 256 384 512; do
  export GFSHIP_FLOW_WIDTH=$W
  echo "== quadtree width $W"
  timeout -k 10 200 $R/tools/lab/tree_prof.sh q$W 2>&1 | grep "relax launches"
  grep -o "'ms_per_step': [0-9.]*" $R/gpurun_out/tree_prof_q$W/log
done
