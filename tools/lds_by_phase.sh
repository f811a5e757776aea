#!/bin/bash
# Diagnostic (item "LDS bank conflicts by phase"): builds of the step kernel that stop at successive phase boundaries, one rocprofv3
# PMC pass each (SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE, SQ_INSTS_LDS, SQ_ACTIVE_INST_LDS of truss_step_kernel, 4096 envs);
# differences between consecutive rows = the phase in between.   build here:  tools/lds_by_phase.sh build ;  run on the GPU box:  tools/lds_by_phase.sh run
cd "$(dirname "$0")/.."
# boundary index (TRUSS_ST) : what has run when the kernel returns there
PH="1:stage 2:decode 3:sizing 10:elements 11:assemble_nodes 4:scratch_init 13:factor_clean_blocks 5:factorisation 6:back_substitution 7:post_elements 8:post_nodes 12:finish 99:whole_step"
if [ "$1" = build ]; then
  for p in $PH; do i=${p%%:*}; tools/abbuild.sh stop$i -DTRUSS_STOP_AT=$i > /dev/null 2>&1 & done; wait; ls mop-truss-marl_amd/csrc/abl/ | grep stop | tr '\n' ' '; exit 0
fi
export TMPDIR=/tmp
O=gpurun_out/lds_by_phase; rm -rf $O; mkdir -p $O
for p in $PH; do
  i=${p%%:*}; n=${p##*:}
  timeout -k 5 60 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $O/p$i -o pmc --output-format csv -- \
    python3 tools/step_only.py mop-truss-marl_amd/csrc/abl/libtruss_stop$i.so 20 > /dev/null 2> $O/p$i.err || { echo "$n failed"; tail -2 $O/p$i.err; exit 1; }
  python3 - $O/p$i $n <<'PY'
import csv, glob, sys, collections, statistics
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "truss_step_kernel" in r["Kernel_Name"] and "false" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: statistics.median(v) for k, v in d.items()}
print(f"{sys.argv[2]:22s} " + "  ".join(f"{k}={m.get(k, float('nan')):.0f}" for k in ("SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT")))
PY
done | tee $O/summary.txt
find $O -name "*.csv" -size +1M -delete
