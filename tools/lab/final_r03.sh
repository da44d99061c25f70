#!/bin/bash
# the whole GPU suite, then the profile recipe of the round
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
mkdir -p $O
cd $R && timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1
rc=$?
echo "tests rc=$rc"; tail -6 $O/tests.log | cut -c 1-200
[ $rc = 0 ] || exit $rc
bash tools/lab/prof_r03.sh
