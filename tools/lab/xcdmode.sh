#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
q='import sys,json; r=json.loads(sys.stdin.read())["roofline"]; print("%.4f %.4f" % (r["ms_per_launch"], r["inclusive"]["ms_per_loop"]))'
for i in 1 2; do
for m in off 0 1 2; do
  if [ $m = off ]; then export GFSHIP_NO_XCD_SCOPE=1; else unset GFSHIP_NO_XCD_SCOPE; export GFSHIP_XCD_NEAR_MODE=$m; fi
  echo "mode $m: $(timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --particles 0 | python3 -c "$q")"
done
done
