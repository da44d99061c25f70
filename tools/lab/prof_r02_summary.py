#!/usr/bin/env python3
"""gpurun_out/prof_r02/ (tools/lab/prof_r02.sh) -> profiles/r02_*: the per-kernel table of the bench
command, the relax loops by level, HBM bytes of the relax kernel from the PMC passes (raw and with
the gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE x 2 for wide coalesced reads), and roofline
entries of the kernels VERDICT r01 asked for (advection, particles) from their average durations."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r02")
DST = os.path.join(ROOT, "profiles")
N = 256
CELLS = N ** 3
PEAK = 8000.0


def find(pattern):
    m = glob.glob(os.path.join(SRC, pattern), recursive=True)
    if not m:
        sys.exit("missing %s" % pattern)
    return m[0]


stats = find("trace/**/b_kernel_stats.csv")
trace = find("trace/**/b_kernel_trace.csv")
shutil.copy(stats, os.path.join(DST, "r02_kernel_stats_bench_256.csv"))
line = open(os.path.join(SRC, "bench.json")).read().strip().splitlines()[-1]
json.loads(line)
open(os.path.join(DST, "r02_bench_line_256.json"), "w").write(line + "\n")

rows = {r["Name"]: r for r in csv.DictReader(open(stats))}


def avg_us(prefix):
    for name, r in rows.items():
        if prefix in name:
            return float(r["AverageNs"]) / 1e3, int(r["Calls"]), name
    return None, 0, None


# relax loops by level from the trace (grid size = tiles x 384 threads)
# threads per tile: six-wave kernel 384, ring kernel (2 x 2 lines per lane + stream wave) 256,
# register variant of it 192
LOOPS = {"relax_skew_loop_kernel": 384, "relax_ring_loop_kernel": 256, "relax_patch_loop_kernel": 192}


def loop_kernel(name):
    for k in LOOPS:
        if k in name:
            return k
    return None


by_level = {}
for r in csv.DictReader(open(trace)):
    k = loop_kernel(r["Kernel_Name"])
    if k is None:
        continue
    grid = int(r["Grid_Size_X"])
    tiles = grid // LOOPS[k]
    n = int(round(tiles ** 0.5)) * 16
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    by_level.setdefault((n, k), []).append(d)
lev = {}
for (n, k), ds in sorted(by_level.items()):
    ds.sort()
    lev["level_n%d" % n] = {"kernel": k, "tiles": (n // 16) ** 2, "dispatches": len(ds),
                            "median_us": ds[len(ds) // 2], "min_us": ds[0], "max_us": ds[-1]}
json.dump({"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 "
                      "--warmup 1 --no-cpu-baseline",
           "note": "dispatches of the sweep-loop kernels in the kernel trace grouped by level (ring kernel "
                   "for n >= 128, six-wave kernel below); at 256^3 the 4-sweep loops of the steps and of "
                   "the roofline entry plus single-sweep launches (hence min << median)",
           "relax_loops": lev}, open(os.path.join(DST, "r02_relax_loop_by_level_from_trace.json"), "w"),
          indent=1)

# PMC: FETCH_SIZE / WRITE_SIZE (KB) of relax_skew_loop_kernel over tools/relax_only.py 8
pm = {}
for key, pat in (("FETCH_SIZE", "pmc_fetch/**/f_counter_collection.csv"),
                 ("WRITE_SIZE", "pmc_write/**/w_counter_collection.csv")):
    f = find(pat)
    shutil.copy(f, os.path.join(DST, "r02_pmc_%s_relax_loop_256.csv" % key))
    tot, launches = 0., 0
    for r in csv.DictReader(open(f)):
        if loop_kernel(r["Kernel_Name"]) and r["Counter_Name"] == key:
            pmc_kernel = loop_kernel(r["Kernel_Name"])
            tot += float(r["Counter_Value"])
            launches += 1
    pm[key] = (tot, launches)
# relax_only.py: time_relax (1 warm-up + 5 single sweeps) + loops of 2 and 4 (warm-up + 5 each) fused,
# then per-sweep mode: (1 + 5) x 2 and (1 + 5) x 4 single sweeps
sweeps = 6 + 6 * 2 + 6 * 4 + 6 * 2 + 6 * 4
fetch_kb, write_kb = pm["FETCH_SIZE"][0] / sweeps, pm["WRITE_SIZE"][0] / sweeps
raw = (fetch_kb + write_kb) * 1024
corr = (2 * fetch_kb + write_kb) * 1024
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 "
                      "tools/relax_only.py 8 (separate passes)",
           "note": "level 8 (256^3); counters summed over the %d launches of the sweep-loop kernel (= %d "
                   "sweeps) and divided by the sweeps; correction per MI355X_MICROARCH.md (FETCH_SIZE x 2 "
                   "for wide coalesced reads: the ring kernel streams its rows with 16-byte LDS-DMA loads; "
                   "the raw sum is kept beside it)" % (pm["FETCH_SIZE"][1], sweeps),
           "kernels": {pmc_kernel: {
               "launches": pm["FETCH_SIZE"][1], "sweeps": sweeps,
               "FETCH_SIZE_KB_per_sweep": fetch_kb, "WRITE_SIZE_KB_per_sweep": write_kb,
               "hbm_bytes_per_sweep_raw": raw, "hbm_bytes_per_sweep_guide_corrected": corr,
               "hbm_bytes_per_launch_raw": 4 * raw, "hbm_bytes_per_launch_guide_corrected": 4 * corr,
               "algorithmic_bytes_per_launch": 24 * CELLS * 4}}},
          open(os.path.join(DST, "r02_pmc_relax_loop_256.json"), "w"), indent=1)

# roofline entries of other kernels from their average durations in the trace
bench = json.loads(line)
entries = {}
for key, prefix, bytes_, what in (
        ("advect3_tiled_kernel", "advect3_tiled_kernel", 120 * CELLS,
         "U, V, W advected in one launch: per cell read v (3 x 8), un (3 x 8), gmac (3 x 8), g (3 x 8), write "
         "out (3 x 8) = 120 B"),
        ("predict_un_tiled_kernel", "predict_un_tiled_kernel", 48 * CELLS, "read u (3 x 8), write un (3 x 8)"),
        ("residual_norm_kernel", "residual_norm_kernel", 32 * CELLS, "read u, rhs, dia, write res"),
        ("project_correct_kernel<3, true>", "project_correct_kernel<3, true>", 128 * CELLS,
         "approximate projection update: DESIGN.md 4"),
        ("patch_pack_kernel", "patch_pack_kernel", None,
         "copy into the skewed layout of the 2 x 2 levels (u by prolongation inside the V-cycle, u and rhs "
         "in the roofline entry), 256^3 and 128^3 mixed"),
        ("patch_restrict_pack_kernel", "patch_restrict_pack_kernel", None,
         "restriction of the residual + its copy into the skewed layout, 256^3 and 128^3 mixed"),
        ("patch_unpack_kernel", "patch_unpack_kernel", None, "copy back (with the correction u += dp at 256^3)"),
        ("particle_list_event_kernel", "particle_list_event_kernel", (1296 + 48) * 2000000,
         "2e6 tracers: 2 stages x 27 cells x 3 components x 8 B gathered + 48 B of state per particle-step"),
        ("particulate_list_event_kernel", "particulate_list_event_kernel", None, "five forces; gather-bound")):
    us, calls, name = avg_us(prefix)
    if us is None:
        continue
    e = {"kernel": name, "calls": calls, "avg_us": us, "what": what}
    if bytes_:
        e.update({"algorithmic_bytes_per_launch": bytes_, "achieved_GBps": bytes_ / (us * 1e-6) / 1e9,
                  "frac_of_8TBps": bytes_ / (us * 1e-6) / 1e9 / PEAK})
    entries[key] = e
json.dump({"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline",
           "note": "average durations of r02_kernel_stats_bench_256.csv x algorithmic bytes per launch; the "
                   "particle gathers are served mostly by L2 / Infinity Cache (sorted by cell), so their "
                   "`achieved' is not HBM traffic",
           "roofline": entries, "bench_line_roofline": bench.get("roofline")},
          open(os.path.join(DST, "r02_kernel_rooflines.json"), "w"), indent=1)
print("profiles/r02_* written")
