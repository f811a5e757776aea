#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel trace of the MADDPG update's hipGraph replays (tools/update_trace.py).
#   gpurun --timeout 600 -- 'bash tools/update_trace.sh'      -> gpurun_out/update_trace/summary.json
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/update_trace
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 python3 tools/update_trace.py run 200 > $OUT/plain.json 2> $OUT/plain.err || { tail -5 $OUT/plain.err; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/run_a -o kt -- python3 tools/update_trace.py run 20 > $OUT/run_a.json 2> $OUT/run_a.err || { tail -5 $OUT/run_a.err; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/run_b -o kt -- python3 tools/update_trace.py run 120 > $OUT/run_b.json 2> $OUT/run_b.err || { tail -5 $OUT/run_b.err; exit 1; }
python3 tools/update_trace.py reduce $OUT > $OUT/summary.json
cat $OUT/plain.json
head -c 1200 $OUT/summary.json
python3 -c "import json;print(json.dumps(json.load(open('$OUT/summary.json'))['level_kernel_by_grid'],indent=1))"
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*.db" -size +8M -delete
