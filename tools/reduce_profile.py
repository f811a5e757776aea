#!/usr/bin/env python3
"""Reduce the rocprofv3 output of tools/profile_round.sh (rocpd SQLite databases) to one JSON summary:
per-kernel statistics of the --kernel-trace pass, per-dispatch medians of every PMC counter for each step-kernel
instantiation (plain / with the fused observation writer), and the HBM bytes per launch with the gfx950
FETCH_SIZE correction of MI355X_MICROARCH.md.  Also writes <out>/pmc_traffic.json, which bench.py reads for
`roofline.traffic` (tagged with the hash of the kernel source it was measured on)."""
import glob
import hashlib
import json
import os
import sqlite3
import statistics
import sys

out = sys.argv[1]
PMC_STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20     # --steps of the PMC passes (profile_round.sh)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = {"kernel_stats": [], "pmc": {}, "dispatch": {}}
for f in glob.glob(os.path.join(out, "kt", "**", "*.db"), recursive=True):
    c = sqlite3.connect(f)
    for name, in c.execute("select distinct name from kernels where name like '%truss%'"):
        d = [r[0] / 1e3 for r in c.execute("select duration from kernels where name = ?", (name,))]
        total_us = sum(d)
        if "rollout_kernel" in name:      # launches of different step counts (warm-up 40, timed 400): the timed ones, per step
            try:
                K = json.loads([l for l in open(os.path.join(out, "bench_under_trace.json")) if l.startswith("{")][-1])["steps"]
            except Exception:  # noqa: BLE001
                K = 400
            d = [x / K for x in d if x >= 0.5 * max(d)]
            name = name + f"  [per step: launch duration / {K} chained steps]"
        g = c.execute("select grid_x, workgroup_x, lds_size from kernels where name = ? limit 1", (name.split("  [")[0],)).fetchone()
        res["kernel_stats"].append({"name": name, "calls": len(d), "average_us": statistics.mean(d), "median_us": statistics.median(d),
                                    "min_us": min(d), "max_us": max(d), "stdev_us": statistics.pstdev(d),
                                    "grid": g[0], "workgroup": g[1], "lds_bytes": g[2], "total_us": total_us})
    tot = c.execute("select sum(duration) from kernels").fetchone()[0]
    for k in res["kernel_stats"]:
        k["share_of_gpu_time"] = k["total_us"] * 1e3 / tot
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*.db"), recursive=True):
    c = sqlite3.connect(f)
    vals = {}
    q = ("select counter_name, value, kernel_name, grid_size, workgroup_size, lds_block_size, scratch_size, vgpr_count, "
         "accum_vgpr_count, sgpr_count from counters_collection where kernel_name like '%truss_step_kernel%' or kernel_name like '%truss_rollout_kernel%'")
    for cn, v, kn, gs, ws, lds, scr, vg, ag, sg in c.execute(q):
        if gs // ws != 1024:          # the 4096-env launches only (16 lanes per env: 1024 workgroups)
            continue
        vals.setdefault(kn, {}).setdefault(cn, []).append(float(v))
        res["dispatch"].setdefault(kn, {"grid": gs, "workgroup": ws, "lds_bytes": lds, "scratch": scr, "vgpr": vg, "accum_vgpr": ag, "sgpr": sg})
    for kn, d in vals.items():
        for k, v in d.items():
            note = None
            if "rollout_kernel" in kn:
                # one launch of the persistent rollout = K chained steps (K = 2 warm-up / 20 timed in the PMC passes): keep
                # the K = PMC_STEPS launches and report counters PER STEP
                big = [x for x in v if x >= 0.5 * max(v)]
                v = [x / PMC_STEPS for x in big]
                note = f"per step: launches of {PMC_STEPS} chained steps, counter / {PMC_STEPS}"
            res["pmc"].setdefault(kn, {})[k] = {"n_dispatches": len(v), "median": statistics.median(v), "min": min(v), "max": max(v)}
            if note:
                res["pmc"][kn][k]["note"] = note
res["hbm_bytes_per_launch"] = {}
for kn, p in res["pmc"].items():
    if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
        rd = p["FETCH_SIZE"]["median"] * 1024 * 2      # KB -> B, x2: gfx950 tallies 128-B requests at 64 B
        wr = p["WRITE_SIZE"]["median"] * 1024
        res["hbm_bytes_per_launch"][kn] = {"read_corrected": rd, "write": wr, "total": rd + wr}
res["hbm_bytes_note"] = "FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes; FETCH_SIZE x2 (gfx950 correction)"
for name in ("bench_under_trace.json", "bench_plain.json"):
    try:
        line = [l for l in open(os.path.join(out, name)) if l.startswith("{")][-1]
        res[name[:-5]] = json.loads(line)
    except Exception as e:  # noqa: BLE001
        res[name[:-5]] = f"unavailable: {e}"
src = open(os.path.join(root, "mop-truss-marl_amd", "csrc", "truss_body.h"), "rb").read()
pick = lambda test: next((v["total"] for k, v in res["hbm_bytes_per_launch"].items() if test(k)), None)
json.dump({"envs": 4096, "nodes": 32, "elements": 80, "truss_body_sha16": hashlib.sha256(src).hexdigest()[:16],
           "step_kernel_bytes_per_launch": pick(lambda k: "step_kernel" in k and "false" in k),      # one launch of truss_step_kernel
           "rollout_kernel_bytes_per_step": pick(lambda k: "rollout_kernel" in k),                   # persistent rollout: per launch / steps
           "fused_step_kernel_bytes_per_launch": pick(lambda k: "step_kernel" in k and "true" in k),
           "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x 2 (gfx950), median over dispatches"},
          open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
