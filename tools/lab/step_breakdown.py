#!/usr/bin/env python3
"""Per-step kernel shares from a rocprofv3 kernel trace of bench.py: the dispatches between the
starts of the 2nd and the last advection launch, summed by kernel (sweep-loop kernels by grid size).
usage: step_breakdown.py <dir with *_kernel_trace.csv> [name of the once-per-step kernel]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
mark = sys.argv[2] if len(sys.argv) > 2 else "advect3"
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
adv = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
if len(adv) < 3:
    sys.exit("fewer than 3 launches of %s" % mark)
a, b = adv[1], adv[min(len(adv) - 1, 5)]
ns = min(len(adv) - 1, 5) - 1
t0, t1 = int(rows[a]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
print("%d steps, wall %.3f ms/step" % (ns, (t1 - t0) / ns / 1e6))
tot, cnt = collections.Counter(), collections.Counter()
for r in rows[a:b]:
    k = r["Kernel_Name"].split("(")[0][:64]
    if "loop_kernel" in r["Kernel_Name"] or "relax" in r["Kernel_Name"]:
        k += " g%s" % r["Grid_Size_X"]
    tot[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cnt[k] += 1
print("busy %.3f ms/step" % (sum(tot.values()) / ns / 1e6))
for k, v in tot.most_common(40):
    print("%-80s %6.1f/step %8.1f us/step  avg %7.1f us" % (k, cnt[k] / ns, v / ns / 1e3, v / cnt[k] / 1e3))
