#!/bin/bash
# test/reynolds/box (`sh ../reynolds.sh box.gfs 4') on the device: the reference's box.gfs, unmodified,
# through gfship2D at LEVEL = 5, 6, 7 (one extra level inside the square: the refined-tree path);
# compares div5, div6, div7 with the reference's files line by line and prints the effective
# Reynolds numbers next to reynolds.ref.   usage: reynolds_box_rows.sh [outdir]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=${1:-$R/gpurun_out/reynolds_box}
mkdir -p $O && cd $O && rm -f reynolds div5 div6 div7
rc=0
for level in 5 6 7; do
  t0=$(date +%s%N)
  sed "s/LEVEL/$level/g" < $R/tests/golden/reference_inputs/reynolds_box.gfs | \
    $R/gerris-fft-particles_amd/bin/gfship2D - | awk -v m=4 -v level=$level '{
      time = $3; ke = $5; if (time == 0) ke0 = ke; }END{
      a = -log(ke/ke0)/time; nu = a/(4.*(2.*m*3.14159265359)^2); print level " " 1./nu }' >> reynolds || exit 1
  echo "level=$level $(( ($(date +%s%N) - t0)/1000000 )) ms, $(wc -l < div$level) rows"
  if diff div$level $R/tests/golden/reference/reynolds_box_div$level.ref > /dev/null; then
    echo "div$level: identical to div$level.ref"
  else echo "div$level: DIFFERS from div$level.ref"; diff div$level $R/tests/golden/reference/reynolds_box_div$level.ref | head -6; rc=1; fi
done
echo "--- reynolds (this run) / reynolds.ref ---"
paste reynolds $R/tests/golden/reference/reynolds_box_reynolds.ref
exit $rc
