#!/bin/bash
# Diagnostic: build stamped single-variant libraries with one ingredient of the factorisation step
# removed each (TRUSS_ABL bit), to be timed with tools/phase_stamps.py <B> <lib>.  Results of ablated
# builds are wrong by construction; only their phase timings are of interest.
set -e
cd "$(dirname "$0")/../mop-truss-marl_amd/csrc"
mkdir -p abl
for m in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -munsafe-fp-atomics \
    -DTRUSS_STAMPS -DTRUSS_ONLY_DEFAULT_VARIANT -DTRUSS_ABL=$m -o abl/libtruss_abl_$m.so truss_hip.hip &
done
wait
ls -la abl
