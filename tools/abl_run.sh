#!/bin/bash
# run on the GPU box: phase timings of every ablated build under mop-truss-marl_amd/csrc/abl
for f in mop-truss-marl_amd/csrc/abl/libtruss_abl_*.so; do timeout -k 10 60 python tools/phase_stamps.py ${1:-4096} $f 2>&1 | grep total || exit 1; done
