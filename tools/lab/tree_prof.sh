#!/bin/bash
# lab: kernel trace of the octree bench (tools/tree_bench.py 3 4 2 5): per-kernel stats and the duration of
# every relax-loop launch in order (which tree level costs what)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/tree_prof${1:+_$1}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python3 $R/tools/tree_bench.py ${TREE_BENCH_ARGS:-3 4 2 3} > $O/log 2>&1
f=$(find $O -name "t_kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print("%-56s calls %5s avg %9.1f us  total %8.2f ms %5.1f %%" % (r["Name"][:56], r["Calls"],
          float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
t=$(find $O -name "t_kernel_trace.csv" | head -1)
python3 - "$t" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "t_relax" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
import os
n = int(os.environ.get("TREE_BENCH_LOOPS", "14"))
print("relax launches:", len(d), "last %d (one step) us:" % n, [round(x, 1) for x in d[-n:]], "sum %.0f" % sum(d[-n:]))
PY
tail -2 $O/log
