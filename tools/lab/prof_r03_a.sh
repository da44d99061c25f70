#!/bin/bash
# round 3, first GPU call: the new N-rank transport tests, the counter list of this box, the kernel
# trace of the bench command (baseline of the round) and the PMC passes of advect3_tiled_kernel
# (FETCH_SIZE, WRITE_SIZE, SQ occupancy / busy / wait counters: each in its own run, program after --)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03a
mkdir -p $O
cd $R && timeout -k 10 900 python -m pytest tests/test_gpu_mock_rccl.py tests/test_gpu_two_ranks.py -m gpu -x -q > $O/tests.log 2>&1
echo "tests rc=$?" ; tail -5 $O/tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
B="python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --particles 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o b -- $B > $O/trace.log 2>&1 && \
rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex advect3 --output-format csv -d $O/pmc_fetch -o f -- $B > $O/pmc_fetch.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --kernel-include-regex advect3 --output-format csv -d $O/pmc_write -o w -- $B > $O/pmc_write.log 2>&1 && \
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --kernel-include-regex advect3 --output-format csv -d $O/pmc_sq -o s -- $B > $O/pmc_sq.log 2>&1
echo "prof rc=$?"
python3 $R/tools/lab/step_breakdown.py $O/trace > $O/step_breakdown.txt 2>&1; head -45 $O/step_breakdown.txt
cd $R && timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.json
