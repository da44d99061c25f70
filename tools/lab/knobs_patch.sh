#!/bin/bash
# fused 4-sweep loop at 256^3 and per-tile time of tile (0,0) (its first sweep waits on nobody) for
# libgfship variants built with tools/build_variant.sh (PK_D, PK_DH, PK_KO ...)
for v in "" "$@"; do
  export GFSHIP_LIB=${GRAFT_REPO_ROOT:-/root/repo}/gerris-fft-particles_amd/lib/libgfship$v.so
  echo "== variant '$v'"
  GFSHIP_SKEW_STATS=1 timeout -k 10 100 python tools/relax_only.py 8 2>&1 | grep -E "tile \( 0, 0\)|tile \(15,15\)|nrelax 4 \(fused" | tail -3 | cut -c1-120
done
