#!/bin/bash
# lab: the 4-sweep relax loop of the levels 5..8 (32^3 .. 256^3) under a few environment settings
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run () {
  echo "== $*"
  for l in 5 6 7 8; do
    env "$@" python3 tools/relax_only.py $l 2>&1 | grep "nrelax 4 (fused" | sed "s/^/L$l /"
  done
}
run A=0
run GFSHIP_XCD_SCOPE=1
run GFSHIP_XCD_SCOPE=1 GFSHIP_XCD_NEAR_MODE=1
run GFSHIP_XCD_PLACE=1
run GFSHIP_PATCH_MIN_N=64
run GFSHIP_PATCH_MIN_N=64 GFSHIP_XCD_SCOPE=1
