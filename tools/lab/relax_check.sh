#!/bin/bash
# lab: the Poisson parity tests, then the relax loop of variants
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-rc}; shift
mkdir -p $O
cd $R && timeout -k 10 900 python -m pytest tests/test_gpu_poisson.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not viscous and not config_d and not taylor_green_step" > $O/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.log | cut -c 1-200
[ $rc = 0 ] || exit $rc
bash tools/lab/relax_variants.sh "$@"
