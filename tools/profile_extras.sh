#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel trace of the secondary kernels -- observation kernel by size class (tools/obs_probe.py), GCN
# aggregation kernels by graph size (tools/agg_probe.py), front / hypervolume + fused step inside batched-rollout game steps
# (tools/marl_bench.py, no training).  Summary: gpurun_out/prof_x/summary.json (copy into profiles/rN/).
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_x
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt1 -o kt -- python3 tools/obs_probe.py > $OUT/obs_probe.json 2> $OUT/kt1.err || { tail -3 $OUT/kt1.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt2 -o kt -- python3 tools/agg_probe.py > $OUT/agg_probe.json 2> $OUT/kt2.err || { tail -3 $OUT/kt2.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt3 -o kt -- python3 tools/marl_bench.py 4096 3 0 > $OUT/marl.json 2> $OUT/kt3.err || { tail -3 $OUT/kt3.err; exit 1; }
python3 - <<'PY' > $OUT/summary.json
import glob, json, sqlite3, statistics
res = {}
for tag in ("kt1", "kt2", "kt3"):
    for f in glob.glob(f"gpurun_out/prof_x/{tag}/**/*.db", recursive=True):
        c = sqlite3.connect(f)
        q = "select name, grid_x, grid_y, workgroup_x, lds_size, count(*), avg(duration), min(duration), max(duration) from kernels where name like '%truss%' group by name, grid_x, grid_y order by name, grid_x"
        for name, gx, gy, wg, lds, n, avg, mn, mx in c.execute(q):
            res.setdefault(tag, []).append({"name": name[:100], "grid": [gx, gy], "workgroup": wg, "lds_bytes": lds, "calls": n,
                                            "average_us": avg / 1e3, "min_us": mn / 1e3, "max_us": mx / 1e3})
def line(p):
    try:
        return json.loads([l for l in open(p) if l.startswith("{")][-1])
    except Exception as e:
        return str(e)
print(json.dumps({"observation_kernel_by_class (tools/obs_probe.py)": res.get("kt1"), "obs_probe_line": line("gpurun_out/prof_x/obs_probe.json"),
                  "aggregation_kernels_by_graph_size (tools/agg_probe.py)": res.get("kt2"), "agg_probe_line": line("gpurun_out/prof_x/agg_probe.json"),
                  "batched_rollout_game_steps (tools/marl_bench.py 4096 3 0)": res.get("kt3"),
                  "note": "rocprofv3 --kernel-trace, one row per (kernel, grid); grid = threads (x), blocks or threads (y) as rocprofv3 reports them"}, indent=1))
PY
head -c 1500 $OUT/summary.json
find $OUT -name "*.db" -size +4M -delete
