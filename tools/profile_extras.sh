#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel trace of the secondary kernels (observation, front/hypervolume, GCN
# aggregation) through bench.py --extras and one batched-rollout game step.  Summary: gpurun_out/prof_x/summary.json
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_x
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt1 -o kt -- python3 bench.py --extras --no-cpu-baseline --steps 100 --warmup 10 > $OUT/bench.json 2> $OUT/kt1.err || { tail -3 $OUT/kt1.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt2 -o kt -- python3 tools/marl_bench.py 4096 3 0 > $OUT/marl.json 2> $OUT/kt2.err || { tail -3 $OUT/kt2.err; exit 1; }
python3 - <<'PY' > $OUT/summary.json
import glob, json, sqlite3, statistics
res = {}
for tag in ("kt1", "kt2"):
    for f in glob.glob(f"gpurun_out/prof_x/{tag}/**/*.db", recursive=True):
        c = sqlite3.connect(f)
        for (name,) in c.execute("select distinct name from kernels where name like '%truss%'"):
            d = [r[0] / 1e3 for r in c.execute("select duration from kernels where name = ?", (name,))]
            g = c.execute("select grid_x, workgroup_x, lds_size from kernels where name = ? limit 1", (name,)).fetchone()
            res.setdefault(tag, []).append({"name": name, "calls": len(d), "average_us": statistics.mean(d), "median_us": statistics.median(d),
                                            "min_us": min(d), "max_us": max(d), "grid": g[0], "workgroup": g[1], "lds_bytes": g[2]})
print(json.dumps({"bench_extras": res.get("kt1"), "marl_game_steps": res.get("kt2"),
                  "note": "rocprofv3 --kernel-trace; kt1 = bench.py --extras (4096 envs), kt2 = tools/marl_bench.py 4096 3 0"}, indent=1))
PY
cat $OUT/summary.json | head -80
find $OUT -name "*.db" -size +4M -delete
