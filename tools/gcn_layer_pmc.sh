#!/bin/bash
# Run ON THE GPU BOX: PMC passes of the bf16x3 GCN layer kernel at 98 304 rows, 200 -> 200 (tools/gcn_layer_one.py 32 200 200).
# -> gpurun_out/gcn_pmc/summary.txt  (counters per dispatch, median)
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/gcn_pmc
rm -rf $OUT; mkdir -p $OUT
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $c | tr ' ' '+')
  timeout -k 10 200 rocprofv3 --pmc $c -d $OUT/$tag -o pmc -- python3 tools/gcn_layer_one.py 32 200 200 > /dev/null 2> $OUT/$tag.err || { echo "pass $c failed"; tail -3 $OUT/$tag.err; }
done
python3 - <<'PY' > $OUT/summary.txt
import glob, sqlite3, statistics
vals = {}
for f in glob.glob("gpurun_out/gcn_pmc/**/*.db", recursive=True):
    c = sqlite3.connect(f)
    try:
        for cn, v, kn in c.execute("select counter_name, value, kernel_name from counters_collection where kernel_name like '%bf3%'"):
            vals.setdefault(cn, []).append(float(v))
    except Exception as e:
        print("?", f, e)
for k in sorted(vals):
    print(f"{k:28s} median {statistics.median(vals[k]):14.0f}   n {len(vals[k])}")
PY
cat $OUT/summary.txt
find $OUT -name "*.db" -size +4M -delete
