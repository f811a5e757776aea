#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py (plain step kernel and
# the fused state-emitting step both run in a default bench.py invocation).
# Summaries land in gpurun_out/prof/ ; copy the ones to be judged into profiles/rN/.
#   gpurun --timeout 900 -- 'bash tools/profile_round.sh'
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
B="python3 bench.py --no-cpu-baseline --no-configs"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- $B --steps 400 --warmup 40 > $OUT/bench_under_trace.json 2> $OUT/kt.err || { tail -5 $OUT/kt.err; exit 1; }
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $c | tr ' ' '+')
  timeout -k 10 200 rocprofv3 --pmc $c -d $OUT/pmc_$tag -o pmc -- $B --steps 20 --warmup 2 > /dev/null 2> $OUT/pmc_$tag.err || { echo "pmc pass $c failed"; tail -3 $OUT/pmc_$tag.err; }
done
timeout -k 10 400 python3 bench.py > $OUT/bench_plain.json 2>/dev/null
python3 tools/reduce_profile.py $OUT 20 > $OUT/summary.json
head -c 3000 $OUT/summary.json
# keep only the small files
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*.db" -size +8M -delete
