#!/bin/bash
# test/periodic/periodic.sh on the device: the nine runs (r = 0, 1, 2 extra levels inside the square,
# LEVEL = 5, 6, 7) of the reference's periodic.gfs, unmodified, through gfship2D; writes r0, r1, r2
# the way the script does (awk '{print level " " $7 " " $9}') and compares them with the
# reference's r0.ref, r1.ref, r2.ref to the printed digits.   usage: periodic_rows.sh [outdir]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=${1:-$R/gpurun_out/periodic_rows}
mkdir -p $O && cd $O && rm -f r0 r1 r2 times
for r in 0 1 2; do
  for level in 5 6 7; do
    t0=$(date +%s%N)
    sed "s/LEVEL/$level/g" < $R/tests/golden/reference_inputs/periodic.gfs | sed "s/BOX/$r/g" | \
      $R/gerris-fft-particles_amd/bin/gfship2D - | \
      awk -v level=$level '{ print level " " $7 " " $9 }' >> r$r || exit 1
    echo "r=$r level=$level $(( ($(date +%s%N) - t0)/1000000 )) ms" | tee -a times
  done
done
rc=0
for r in 0 1 2; do
  if diff r$r $R/tests/golden/reference/periodic_r$r.ref > /dev/null; then echo "r$r: identical to r$r.ref"
  else echo "r$r: DIFFERS from r$r.ref"; diff r$r $R/tests/golden/reference/periodic_r$r.ref; rc=1; fi
done
exit $rc
