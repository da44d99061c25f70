#!/usr/bin/env python3
"""gpurun_out/prof_r03/ (tools/lab/prof_r03.sh) + gpurun_out/r03a/ (the counters of the round-2
advection kernel, tools/lab/prof_r03_a.sh) + gpurun_out/pmc_sweep/ -> profiles/r03_*."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r03")
DST = os.path.join(ROOT, "profiles")
N = 256
CELLS = N ** 3
PEAK = 8000.0


def find(pattern, base=SRC, must=True):
    m = glob.glob(os.path.join(base, pattern), recursive=True)
    if not m:
        if must:
            sys.exit("missing %s" % pattern)
        return None
    return m[0]


stats = find("trace/**/b_kernel_stats.csv")
trace = find("trace/**/b_kernel_trace.csv")
shutil.copy(stats, os.path.join(DST, "r03_kernel_stats_bench_256.csv"))
shutil.copy(os.path.join(SRC, "step_breakdown.txt"), os.path.join(DST, "r03_step_breakdown.txt"))
line = open(os.path.join(SRC, "bench.json")).read().strip().splitlines()[-1]
bench = json.loads(line)
open(os.path.join(DST, "r03_bench_line_256.json"), "w").write(line + "\n")
sm = os.path.join(SRC, "selfmpi.json")
if os.path.exists(sm) and open(sm).read().strip():
    open(os.path.join(DST, "r03_selfmpi_bench_line.json"), "w").write(open(sm).read().strip().splitlines()[-1] + "\n")
pr = os.path.join(SRC, "periodic_rows.txt")
if os.path.exists(pr):
    shutil.copy(pr, os.path.join(DST, "r03_periodic_rows.txt"))

for name in ("weighted_cycle", "tree_bench", "refined_cavity"):
    src = os.path.join(SRC, name + ".txt")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(DST, "r03_%s.txt" % name))

rows = {r["Name"]: r for r in csv.DictReader(open(stats))}


def avg_us(prefix):
    for name, r in rows.items():
        if prefix in name:
            return float(r["AverageNs"]) / 1e3, int(r["Calls"]), name
    return None, 0, None


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
    tiles = int(r["Grid_Size_X"]) // LOOPS[k]
    n = int(round(tiles ** 0.5)) * 16
    by_level.setdefault((n, k), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
lev = {}
for (n, k), ds in sorted(by_level.items()):
    ds.sort()
    lev["level_n%d" % n] = {"kernel": k, "tiles": (n // 16) ** 2, "dispatches": len(ds),
                            "median_us": ds[len(ds) // 2], "min_us": ds[0], "max_us": ds[-1]}
json.dump({"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 "
                      "--warmup 1 --no-cpu-baseline",
           "note": "dispatches of the sweep-loop kernels in the kernel trace grouped by level; at 256^3 the "
                   "4-sweep loops of the steps and of the roofline entry plus single-sweep launches",
           "relax_loops": lev}, open(os.path.join(DST, "r03_relax_loop_by_level_from_trace.json"), "w"), indent=1)

# PMC of the relax loop: as round 2
pm = {}
pmc_kernel = None
for key, pat in (("FETCH_SIZE", "pmc_fetch/**/f_counter_collection.csv"),
                 ("WRITE_SIZE", "pmc_write/**/w_counter_collection.csv")):
    f = find(pat)
    shutil.copy(f, os.path.join(DST, "r03_pmc_%s_relax_loop_256.csv" % key))
    tot, launches = 0., 0
    for r in csv.DictReader(open(f)):
        if loop_kernel(r["Kernel_Name"]) and r["Counter_Name"] == key:
            pmc_kernel = loop_kernel(r["Kernel_Name"])
            tot += float(r["Counter_Value"])
            launches += 1
    pm[key] = (tot, launches)
sweeps = 6 + 6 * 2 + 6 * 4 + 6 * 2 + 6 * 4      # tools/relax_only.py, see prof_r02_summary.py
fetch_kb, write_kb = pm["FETCH_SIZE"][0] / sweeps, pm["WRITE_SIZE"][0] / sweeps
raw = (fetch_kb + write_kb) * 1024
corr = (2 * fetch_kb + write_kb) * 1024
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 "
                      "tools/relax_only.py 8 (separate passes)",
           "note": "level 8 (256^3); counters summed over the %d launches of the sweep-loop kernel (= %d sweeps) "
                   "and divided by the sweeps; correction per MI355X_MICROARCH.md (FETCH_SIZE x 2 for wide "
                   "coalesced reads: the ring kernel streams its rows with 16-byte LDS-DMA loads; the raw sum "
                   "is kept beside it)" % (pm["FETCH_SIZE"][1], sweeps),
           "kernels": {pmc_kernel: {
               "launches": pm["FETCH_SIZE"][1], "sweeps": sweeps,
               "FETCH_SIZE_KB_per_sweep": fetch_kb, "WRITE_SIZE_KB_per_sweep": write_kb,
               "hbm_bytes_per_sweep_raw": raw, "hbm_bytes_per_sweep_guide_corrected": corr,
               "hbm_bytes_per_launch_raw": 4 * raw, "hbm_bytes_per_launch_guide_corrected": 4 * corr,
               "algorithmic_bytes_per_launch": 24 * CELLS * 4}}},
          open(os.path.join(DST, "r03_pmc_relax_loop_256.json"), "w"), indent=1)


# counters of the advection kernels: the round-2 tiled kernel (gpurun_out/r03a) and the sweep kernels
def counters(base, sub):
    out = {}
    for f in glob.glob(os.path.join(base, sub, "**", "*counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"].split("(")[0].split("::")[-1], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            out.setdefault(k, {})[c] = sum(v) / len(v)
    return out


def digest(c, ms, bytes_alg):
    d = dict(c)
    w = c.get("SQ_WAVES")
    if w and c.get("SQ_WAVE_CYCLES"):
        d["derived"] = {
            "valu_instructions_per_wave": c["SQ_INSTS_VALU"] / w,
            "wait_any_share_of_wave_cycles": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
            "wait_inst_any_share": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
            "active_inst_any_share": c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
            # SQ_ACTIVE_INST_VALU counts quad-cycles; 1024 SIMDs; clock from the duration is not known here:
            "valu_busy_quad_cycles_per_simd": c["SQ_ACTIVE_INST_VALU"] / 1024.,
        }
    if "FETCH_SIZE" in c:
        d["hbm_bytes_raw"] = (c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0.)) * 1024
        d["hbm_bytes_fetch_x2"] = (2 * c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0.)) * 1024
        d["algorithmic_bytes"] = bytes_alg
        d["traffic_over_algorithmic_raw"] = d["hbm_bytes_raw"] / bytes_alg
    if ms:
        d["avg_ms"] = ms
    return d


adv = {}
old = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for k, c in counters(os.path.join(ROOT, "gpurun_out", "r03a"), sub).items():
        old.setdefault(k, {}).update(c)
for k, c in old.items():
    if "advect3_tiled" in k:
        adv["advect3_tiled_kernel (round 2 kernel, start of round 3)"] = digest(c, 1.139, 120 * CELLS)
new = {}
for sub in ("fetch", "write", "sq1", "sq2"):
    for k, c in counters(os.path.join(ROOT, "gpurun_out", "pmc_sweep"), sub).items():
        new.setdefault(k, {}).update(c)
for k, c in new.items():
    us, _, _ = avg_us(k.split("<")[0])
    alg = (120 + 72) * CELLS if "advect3" in k else (48 + 8) * CELLS
    adv[k] = digest(c, us / 1e3 if us else None, alg)
json.dump({"command": "rocprofv3 --pmc <group> --kernel-trace --kernel-include-regex <kernel> -- python3 bench.py "
                      "--steps 3|4 --warmup 1 --no-cpu-baseline --particles 0; one run per group (FETCH_SIZE; "
                      "WRITE_SIZE; SQ waves / wait / VALU; SQ LDS / VMEM): tools/lab/prof_r03_a.sh, "
                      "tools/lab/pmc_kernel.sh",
           "note": "per launch at 256^3, averages over the dispatches of the run. FETCH_SIZE / WRITE_SIZE in KB "
                   "as rocprofv3 reports them; SQ_*CYCLES and SQ_ACTIVE / SQ_WAIT in quad-cycles summed over the "
                   "waves (MI355X_MICROARCH.md). The loads of these kernels are 8 bytes per lane: the guide's "
                   "x2 on FETCH_SIZE is calibrated for 16-byte streams only, so the raw sum is the one "
                   "compared with the algorithmic bytes, the x2 variant is given beside it. Algorithmic bytes "
                   "of the sweep kernels include the passes fused into them (advection: + centred correction "
                   "48 + 24 B per cell; predictor: + the divergence written, 8 B per cell).",
           "kernels": adv}, open(os.path.join(DST, "r03_pmc_advect3.json"), "w"), indent=1)

# roofline entries of other kernels from their average durations in the trace
entries = {}
for key, prefix, bytes_, what in (
        ("advect3_sweep_kernel", "advect3_sweep_kernel", (120 + 72) * CELLS,
         "U, V, W advected + gfs_correct_centered_velocities + first coarse level in one sweep along z: read v, "
         "un, gmac, g (12 x 8), write u (3 x 8) = 120 B, plus the 72 B of the fused correction pass (read u, g, "
         "write u) that no longer exists"),
        ("advect3_sweep_kernel_on_120B", "advect3_sweep_kernel", 120 * CELLS,
         "the same duration on the bytes of the unfused advection alone (round 2's definition)"),
        ("predict_un_sweep_kernel", "predict_un_sweep_kernel", (48 + 8) * CELLS,
         "read u (3 x 8), write un (3 x 8), write div (8)"),
        ("residual_norm2_kernel", "residual_norm2_kernel<true>", 24 * CELLS, "read u, rhs, write res (dia == 0 not read)"),
        ("project_correct_kernel<3, true>", "project_correct_kernel<3, true>", 128 * CELLS,
         "approximate projection update: DESIGN.md 4"),
        ("project_correct_kernel<3, false>", "project_correct_kernel<3, false>", 80 * CELLS, "MAC projection update"),
        ("face_interp_div_kernel", "face_interp_div_kernel", 56 * CELLS, "read u (3 x 8), write un (3 x 8), div"),
        ("patch_prolong_kernel", "patch_prolong_kernel", None,
         "get_from_above straight into the patch layout (256^3 and 128^3 mixed)"),
        ("patch_restrict_pack_kernel", "patch_restrict_pack_kernel", None,
         "get_from_below + the residual as the loop's rhs (256^3 and 128^3 mixed)"),
        ("patch_unpack_kernel", "patch_unpack_kernel", None,
         "out of the layout (with correct, u += dp, at 256^3); 256^3 and 128^3 mixed, plus the roofline entry's"),
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
           "note": "average durations of r03_kernel_stats_bench_256.csv x algorithmic bytes per launch; the "
                   "particle gathers are served mostly by L2 / Infinity Cache (sorted by cell), so their "
                   "`achieved' is not HBM traffic",
           "roofline": entries, "bench_line_roofline": bench.get("roofline")},
          open(os.path.join(DST, "r03_kernel_rooflines.json"), "w"), indent=1)
print("profiles/r03_* written")
