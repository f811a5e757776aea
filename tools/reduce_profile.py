#!/usr/bin/env python3
"""Reduce the rocprofv3 output of tools/profile_round.sh (rocpd SQLite databases) to one JSON summary:
per-kernel statistics of the --kernel-trace pass, per-dispatch medians of every PMC counter for the step
kernel, and the HBM bytes per launch with the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md."""
import glob
import json
import os
import sqlite3
import statistics
import sys

out = sys.argv[1]
res = {"kernel_stats": [], "pmc": {}}
for f in glob.glob(os.path.join(out, "kt", "**", "*.db"), recursive=True):
    c = sqlite3.connect(f)
    for name, in c.execute("select distinct name from kernels where name like '%truss%'"):
        d = [r[0] / 1e3 for r in c.execute("select duration from kernels where name = ?", (name,))]
        res["kernel_stats"].append({"name": name, "calls": len(d), "average_us": statistics.mean(d), "median_us": statistics.median(d),
                                    "min_us": min(d), "max_us": max(d), "stdev_us": statistics.pstdev(d)})
    tot = c.execute("select sum(duration) from kernels").fetchone()[0]
    for k in res["kernel_stats"]:
        k["share_of_gpu_time"] = k["average_us"] * k["calls"] * 1e3 / tot
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*.db"), recursive=True):
    c = sqlite3.connect(f)
    vals = {}
    q = ("select counter_name, value, kernel_name, grid_size, workgroup_size, lds_block_size, scratch_size, vgpr_count, "
         "accum_vgpr_count, sgpr_count from counters_collection where kernel_name like '%truss_step_kernel%'")
    for cn, v, kn, gs, ws, lds, scr, vg, ag, sg in c.execute(q):
        vals.setdefault(cn, []).append(float(v))
        res.setdefault("dispatch", {"kernel": kn, "grid": gs, "workgroup": ws, "lds_bytes": lds, "scratch": scr, "vgpr": vg,
                                    "accum_vgpr": ag, "sgpr": sg})
    for k, v in vals.items():
        res["pmc"][k] = {"n_dispatches": len(v), "median": statistics.median(v), "min": min(v), "max": max(v)}
p = res["pmc"]
if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
    rd = p["FETCH_SIZE"]["median"] * 1024 * 2      # KB -> B, x2: gfx950 tallies 128-B requests at 64 B
    wr = p["WRITE_SIZE"]["median"] * 1024
    res["hbm_bytes_per_launch"] = {"read_corrected": rd, "write": wr, "total": rd + wr,
                                   "note": "FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes; FETCH_SIZE x2 (gfx950 correction)"}
for name in ("bench_under_trace.json", "bench_plain.json"):
    try:
        line = [l for l in open(os.path.join(out, name)) if l.startswith("{")][-1]
        res[name[:-5]] = json.loads(line)
    except Exception as e:  # noqa: BLE001
        res[name[:-5]] = f"unavailable: {e}"
print(json.dumps(res, indent=1))
