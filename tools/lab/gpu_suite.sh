#!/bin/bash
# the whole GPU suite, then a profiled bench run (per-step kernel table) and the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-suite}
mkdir -p $O
cd $R && timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1
rc=$?
echo "tests rc=$rc"; tail -12 $O/tests.log
[ $rc = 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o b -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --particles 0 > $O/trace.log 2>&1
python3 $R/tools/lab/step_breakdown.py $O/trace > $O/step_breakdown.txt 2>&1; head -34 $O/step_breakdown.txt
cd $R && timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python3 - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.0f ms/step %.3f frac %.3f incl %.3f vcycle %.3f" % (d["value"], d["ms_per_step"], r["frac"], r["inclusive"]["frac"], r["vcycle"]["ms"]))
PY
