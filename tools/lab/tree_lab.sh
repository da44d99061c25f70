#!/bin/bash
# lab: what each part of a level of t_relax_flow costs (FLOW_LAB variants built with tools/build_variant.sh labN "-DFLOW_LAB=N")
R=${GRAFT_REPO_ROOT:-/root/repo}
for W in 512 64; do
export GFSHIP_FLOW_WIDTH=$W
for v in "" lab2 lab6 lab14 lab15; do
  if [ -n "$v" ]; then export GFSHIP_LIB=$R/gerris-fft-particles_amd/lib/libgfship_$v.so; else unset GFSHIP_LIB; fi
  echo "== width $W variant ${v:-default}"
  timeout -k 10 120 $R/tools/lab/tree_prof.sh v$v 2>&1 | grep "relax launches"
done
done
