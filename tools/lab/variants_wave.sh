#!/bin/bash
# like variants.sh, for relax_wave_loop_kernel (GFSHIP_WAVE_LOOP=1)
for v in "" "$@"; do
  export GFSHIP_LIB=${GRAFT_REPO_ROOT:-/root/repo}/gerris-fft-particles_amd/lib/libgfship$v.so
  echo "== variant '$v'"
  GFSHIP_WAVE_LOOP=1 GFSHIP_SKEW_STATS=1 timeout -k 10 100 python tools/relax_only.py 8 2>&1 | grep -E "tile \( 0, 0\)|tile \(15,15\)|nrelax 4 \(fused" | tail -3 | cut -c1-110
done
