#!/bin/bash
# The profile recipe behind profiles/r03_*: run on the GPU box from the repo root
#   (gpurun -- 'bash tools/lab/prof_r03.sh'); outputs under gpurun_out/prof_r03/, and
#   tools/lab/prof_r03_summary.py turns them into the files committed under profiles/.
# 1. kernel trace + stats of the bench command (with the config D particle lines)
# 2./3. FETCH_SIZE / WRITE_SIZE of the relax loop kernel (tools/relax_only.py 8), own passes
# 4. PMC passes of the Godunov sweep kernels (tools/lab/pmc_kernel.sh)
# 5. the bench line itself, un-profiled; the box with MPI sides to itself; the 2-D driver of the reference
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r03
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o b -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/trace.log 2>&1 && \
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 $R/tools/relax_only.py 8 > $O/pmc_fetch.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 $R/tools/relax_only.py 8 > $O/pmc_write.log 2>&1
echo "relax pmc rc=$?"
bash $R/tools/lab/pmc_kernel.sh "advect3_sweep|predict_un_sweep" sweep > $O/pmc_sweep.txt 2>&1
python3 $R/tools/lab/step_breakdown.py $O/trace > $O/step_breakdown.txt 2>&1
cd $R && timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.json
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --self-mpi --no-cpu-baseline --particles 0 > $O/selfmpi.json 2> $O/selfmpi.err
tail -c 300 $O/selfmpi.json
( time bash tools/periodic_rows.sh ) > $O/periodic_rows.txt 2>&1
tail -5 $O/periodic_rows.txt
# 6. the paths added late in round 3: the weighted cycle, the refined-tree bench, the refined lid-driven cavity
( timeout -k 10 120 python tools/weighted_cycle.py 7; timeout -k 10 120 python tools/weighted_cycle.py 8 ) > $O/weighted_cycle.txt 2>&1
( timeout -k 10 200 python tools/tree_bench.py 3 4 2 5; timeout -k 10 200 python tools/tree_bench.py 2 7 2 5 ) > $O/tree_bench.txt 2>&1
( time gerris-fft-particles_amd/bin/gfship2D -DLEVEL=5 -DNSTEPS=100000 tests/cases/refined_cavity.gfs ) > $O/refined_cavity.txt 2>&1
tail -3 $O/weighted_cycle.txt $O/tree_bench.txt $O/refined_cavity.txt
