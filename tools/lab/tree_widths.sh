#!/bin/bash
# lab: the relax kernel of the octree bench for several plan widths (GFSHIP_FLOW_WIDTH): us per launch of the last cycle
R=${GRAFT_REPO_ROOT:-/root/repo}
for W in ${WIDTHS:-64 128 256 384 512}; do
  export GFSHIP_FLOW_WIDTH=$W
  echo "== width $W"
  timeout -k 10 200 $R/tools/lab/tree_prof.sh w$W 2>&1 | grep "relax launches\|ms_per_step"
  grep -o "'ms_per_step': [0-9.]*" $R/gpurun_out/tree_prof_w$W/log
done
