#!/bin/bash
# lab: PMC passes of one kernel (regex $1) of the bench command, each in its own run: FETCH_SIZE,
# WRITE_SIZE, SQ wave / wait counters, SQ LDS / VMEM counters.  Output under gpurun_out/pmc_$2/
R=${GRAFT_REPO_ROOT:-/root/repo}
RE=$1; O=$R/gpurun_out/pmc_${2:-k}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --particles 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "$RE" --output-format csv -d $O/fetch -o f -- $B > $O/fetch.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --kernel-include-regex "$RE" --output-format csv -d $O/write -o w -- $B > $O/write.log 2>&1 && \
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace --kernel-include-regex "$RE" --output-format csv -d $O/sq1 -o s -- $B > $O/sq1.log 2>&1 && \
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --kernel-include-regex "$RE" --output-format csv -d $O/sq2 -o t -- $B > $O/sq2.log 2>&1
echo "pmc rc=$?"
python3 - $O <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + "/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        acc[(r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print("%-62s %-24s n=%3d avg %16.1f" % (k, c, len(v), sum(v) / len(v)))
    if rows:
        print("   VGPR", rows[0].get("VGPR_Count"), "LDS", rows[0].get("LDS_Block_Size"), "WG", rows[0].get("Workgroup_Size"))
PY
