#!/bin/bash
# lab: parity tests that cover the advection sweeps, then the kernels with and without GFSHIP_ADVECT_SWEEP1
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-adv}
mkdir -p $O
cd $R && timeout -k 10 900 python -m pytest tests/test_gpu_timestep.py tests/test_gpu_fullsize.py -m gpu -x -q -k "sweep_kernels or taylor_green_step or fused_periodic" > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -6 $O/tests.log | cut -c 1-300
[ $rc = 0 ] || exit $rc
bash $R/tools/lab/ab_libs.sh "advect3|predict_un" main
export GFSHIP_ADVECT_SWEEP1=1
bash $R/tools/lab/ab_libs.sh "advect3|predict_un" main
