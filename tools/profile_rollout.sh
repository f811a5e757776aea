#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel trace of the batched MADDPG rollout (BASELINE configs[2]: small_roof, 4096 envs),
# without and with training.  Summary: gpurun_out/prof_marl/summary.json (copy into profiles/rN/).
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof_marl
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt0 -o kt -- python3 tools/marl_bench.py 4096 3 0 > $OUT/marl_notrain.json 2> $OUT/kt0.err || { tail -3 $OUT/kt0.err; exit 1; }
# the update's GEMM choices (TunableOp) are made in an untraced run and read back, so that the trace holds no tuning candidates
export TRUSS_GEMM_TUNE_FILE=/tmp/truss_tunableop.csv
timeout -k 10 300 python3 tools/marl_bench.py 4096 1 1 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt1 -o kt -- python3 tools/marl_bench.py 4096 3 1 > $OUT/marl_train.json 2> $OUT/kt1.err || { tail -3 $OUT/kt1.err; exit 1; }
python3 - <<'PY' > $OUT/summary.json
import glob, json, sqlite3, statistics
res = {}
for tag in ("kt0", "kt1"):
    for f in glob.glob(f"gpurun_out/prof_marl/{tag}/**/*.db", recursive=True):
        c = sqlite3.connect(f)
        tot = c.execute("select sum(duration), count(*) from kernels").fetchone()
        rows = c.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name order by sum(duration) desc limit 14").fetchall()
        res[tag] = {"gpu_kernel_time_ms": tot[0] / 1e6, "kernel_launches": tot[1],
                    "top_kernels": [{"name": n[:110], "calls": k, "total_ms": s / 1e6, "avg_us": a / 1e3, "share": s / tot[0]} for n, k, s, a in rows]}
def line(p):
    try:
        return json.loads([l for l in open(p) if l.startswith("{")][-1])
    except Exception as e:
        return str(e)
print(json.dumps({"no_training (warm-up + 3 game steps)": res.get("kt0"), "with_training (warm-up + 3 game steps)": res.get("kt1"),
                  "bench_no_training_under_trace": line("gpurun_out/prof_marl/marl_notrain.json"),
                  "bench_training_under_trace": line("gpurun_out/prof_marl/marl_train.json"),
                  "note": "rocprofv3 --kernel-trace --stats of tools/marl_bench.py 4096 3 {0,1}: small_roof 16 nodes / 36 elements, 4096 envs"}, indent=1))
PY
python3 tools/kernels_by_count.py $OUT/kt1 60 > $OUT/kt1_by_count.txt
python3 tools/kernels_by_count.py $OUT/kt0 30 > $OUT/kt0_by_count.txt
head -c 2500 $OUT/summary.json
find $OUT -name "*.db" -size +4M -delete
