#!/bin/bash
# same-box A/B of the FUSED state-emitting step (tools/fused_probe.py) over builds under csrc/abl:  tools/abfused.sh reps envs name1 name2 ...
reps=$1; envs=$2; shift; shift
for r in $(seq $reps); do for p in "$@"; do echo -n "$p "; timeout -k 10 120 python tools/fused_probe.py $envs mop-truss-marl_amd/csrc/abl/libtruss_$p.so 2>&1 | grep us_per_step || exit 1; done; done
