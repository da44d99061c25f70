#!/bin/bash
# lab: A/B of an environment switch on the same box: bench.py ms_per_step with and without it, alternating
# usage: ab.sh VAR [runs]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
V=$1; N=${2:-3}
for i in $(seq $N); do
  a=$(timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --particles 0 | python3 -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(env $V=1 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --particles 0 | python3 -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
  echo "default $a   $V=1 $b"
done
