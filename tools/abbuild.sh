#!/bin/bash
# build an experiment variant of the HIP library (default kernel variant only):  tools/abbuild.sh <name> [extra hipcc flags...]
# -> mop-truss-marl_amd/csrc/abl/libtruss_<name>.so (git-ignored; travels to the GPU box with gpurun)
set -e
cd "$(dirname "$0")/../mop-truss-marl_amd/csrc"
name=$1; shift
mkdir -p abl
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -munsafe-fp-atomics -Wall -Wno-unused-function \
  -Wno-unknown-pragmas -DTRUSS_ONLY_DEFAULT_VARIANT "$@" -Rpass-analysis=kernel-resource-usage -o abl/libtruss_$name.so truss_hip.hip 2> abl/$name.res
grep -A12 "truss_step_kernel.*Lb1" abl/$name.res | grep -E "VGPRs:|ScratchSize|Occupancy|LDS Size" | head -4 | tr '\n' ' '; echo
