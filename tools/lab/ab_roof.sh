#!/bin/bash
# lab: A/B of an environment switch on the same box: relax loop kernel-only / inclusive ms
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
V=$1; N=${2:-3}
q='import sys,json; r=json.loads(sys.stdin.read())["roofline"]; print("%.4f %.4f" % (r["ms_per_launch"], r["inclusive"]["ms_per_loop"]))'
for i in $(seq $N); do
  a=$(timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --particles 0 | python3 -c "$q")
  b=$(env $V=1 timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --particles 0 | python3 -c "$q")
  echo "default $a   $V=1 $b"
done
