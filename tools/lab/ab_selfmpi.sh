#!/bin/bash
# lab: A/B of an environment switch on the multi-box code path (bench.py --self-mpi), same box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
V=$1; N=${2:-3}
q='import sys,json; print("%.4f" % json.loads(sys.stdin.read().strip().splitlines()[-1])["ms_per_step"])'
for i in $(seq $N); do
  a=$(timeout -k 10 200 python bench.py --self-mpi --steps 8 --warmup 2 --no-cpu-baseline --particles 0 2>/dev/null | python3 -c "$q")
  b=$(env $V=1 timeout -k 10 200 python bench.py --self-mpi --steps 8 --warmup 2 --no-cpu-baseline --particles 0 2>/dev/null | python3 -c "$q")
  echo "default $a   $V=1 $b"
done
