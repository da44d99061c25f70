#!/bin/bash
# loop time at 256^3 (and 128^3) of libgfship variants built with tools/build_variant.sh
for v in "" "$@"; do
  export GFSHIP_LIB=${GRAFT_REPO_ROOT:-/root/repo}/gerris-fft-particles_amd/lib/libgfship$v.so
  echo "== variant '$v'"
  for lev in 8 7; do
    timeout -k 10 100 python tools/relax_only.py $lev 2>&1 | grep -E "ms per sweep|nrelax 4 \(fused" | cut -c1-120
  done
done
