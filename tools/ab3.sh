#!/bin/bash
# same-box comparison of several HIP builds under mop-truss-marl_amd/csrc/abl:  tools/ab3.sh reps name1 name2 ...
reps=$1; shift
for r in $(seq $reps); do for p in "$@"; do echo -n "$p "; timeout -k 10 300 python bench.py --no-cpu-baseline --lib mop-truss-marl_amd/csrc/abl/libtruss_$p.so 2>&1 | grep -o "\"kernel_us\": [0-9.]*" | tr '\n' ' ' || exit 1; echo; done; done
