#!/bin/bash
# lab: per-kernel averages of a profiled bench run for several library variants (tools/build_variant.sh)
# usage: ab_libs.sh REGEX name1 name2 ...   ("main" = libgfship.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
RE=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  O=$R/gpurun_out/ab_$v; rm -rf $O; mkdir -p $O
  if [ "$v" = main ]; then unset GFSHIP_LIB; else export GFSHIP_LIB=$R/gerris-fft-particles_amd/lib/libgfship_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o b -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --particles 0 > $O/log 2>&1
  f=$(find $O -name "b_kernel_stats.csv" | head -1)
  echo "== $v"
  python3 - "$f" "$RE" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print("%-70s calls %4s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  grep '^{"metric"' $O/log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step %.3f' % d['ms_per_step'])"
done
