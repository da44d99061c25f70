#!/bin/bash
# lab: SQ counters of the relax kernel of the octree bench (tools/tree_bench.py 3 4 2 2), per dispatch:
# the largest dispatch (finest level) and a small one (level 3)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/tree_pmc${1:+_$1}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/tools/tree_bench.py 3 4 2 2"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --kernel-include-regex "t_relax" --output-format csv -d $O/sq1 -o s -- $B > $O/sq1.log 2>&1 && \
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --kernel-trace --kernel-include-regex "t_relax" --output-format csv -d $O/sq2 -o t -- $B > $O/sq2.log 2>&1
echo "pmc rc=$?"
python3 - $O <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    by = collections.defaultdict(dict)
    for r in rows:
        by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(by)
    last = ids[-7:]           # the last cycle: levels 0..6
    for i in (last[3], last[5], last[6]):
        print("dispatch", i, " ".join("%s=%.0f" % kv for kv in sorted(by[i].items())))
PY
