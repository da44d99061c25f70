#!/bin/bash
# lab: the refined lid-driven cavity (3328 leaves: every level of its plans is narrow) for several plan widths
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for W in "" 64 128 256; do
  if [ -n "$W" ]; then export GFSHIP_FLOW_WIDTH=$W; else unset GFSHIP_FLOW_WIDTH; fi
  t0=$(date +%s%N)
  gerris-fft-particles_amd/bin/gfship2D -DLEVEL=5 -DNSTEPS=100000 tests/cases/refined_cavity.gfs > /dev/null 2> gpurun_out/cav_$W.err
  echo "width ${W:-default}: $(( ($(date +%s%N) - t0)/1000000 )) ms"
done
