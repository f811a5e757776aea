#!/bin/bash
# same-box A/B of HIP builds under mop-truss-marl_amd/csrc/abl:  tools/ab.sh a b [reps]
# (build them with  hipcc ... -DTRUSS_ONLY_DEFAULT_VARIANT -o abl/libtruss_<name>.so truss_hip.hip)
reps=${3:-2}
for r in $(seq $reps); do for p in $1 $2; do echo -n "$p "; timeout -k 10 300 python bench.py --no-cpu-baseline --lib mop-truss-marl_amd/csrc/abl/libtruss_$p.so 2>&1 | grep -o "\"kernel_us\": [0-9.]*" || exit 1; done; done
