#!/bin/bash
# lab: parity tests that cover the advection kernels, then the per-kernel averages of a bench run
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-adv}
mkdir -p $O
cd $R && timeout -k 10 900 python -m pytest tests/test_gpu_timestep.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not viscous_taylor_green_128" > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -8 $O/tests.log | cut -c 1-300
[ $rc = 0 ] || exit $rc
bash $R/tools/lab/ab_libs.sh "advect3|predict_un|residual|patch_" main
