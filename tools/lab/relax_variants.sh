#!/bin/bash
# lab: 4-sweep relax loop at 256^3 and 128^3 for library variants (tools/build_variant.sh), three repetitions each
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do
for v in main "$@"; do
  if [ "$v" = main ]; then unset GFSHIP_LIB; else export GFSHIP_LIB=$R/gerris-fft-particles_amd/lib/libgfship_$v.so; fi
  a=$(python3 tools/relax_only.py 8 2>&1 | grep "nrelax 4 (fused" | awk '{print $7}')
  b=$(python3 tools/relax_only.py 7 2>&1 | grep "nrelax 4 (fused" | awk '{print $7}')
  echo "rep $rep $v 256: $a  128: $b"
done
done
