#!/bin/bash
# the tree path after the flow kernel: every GPU test that touches a tree, the two tree benches, the reference's
# periodic.sh and reynolds/box drivers (outputs diffed against its golden files) with their wall times
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/tree_final
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_tree.py tests/test_gfs_frontend.py tests/test_reference_files.py tests/test_gpu_snapshot.py -q -m gpu > $O/tests.log 2>&1
tail -3 $O/tests.log
( timeout -k 10 200 python tools/tree_bench.py 3 4 2 5; timeout -k 10 200 python tools/tree_bench.py 2 7 2 5 ) > $O/tree_bench.txt 2>&1
grep "^{" $O/tree_bench.txt
( GFSHIP_TREE_NO_FLOW=1 timeout -k 10 200 python tools/tree_bench.py 3 4 2 5; GFSHIP_TREE_NO_FLOW=1 timeout -k 10 200 python tools/tree_bench.py 2 7 2 5 ) > $O/tree_bench_tapes.txt 2>&1
grep "^{" $O/tree_bench_tapes.txt
timeout -k 10 400 tools/periodic_rows.sh $O/periodic > $O/periodic_rows.txt 2>&1; tail -12 $O/periodic_rows.txt
