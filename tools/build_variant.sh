#!/bin/bash
# Build an A/B variant of libmadsim_hip.so:  tools/build_variant.sh <tag> [extra hipcc flags...]
set -e
tag=$1; shift
cd "$(dirname "$0")/../madsim_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Os -std=c++17 -fPIC -Wall -Wno-unused-function -DMADSIM_EXPERIMENT_BUILD "$@" -c sim_kernel.hip -o /tmp/sim_kernel_$tag.o
# (the host side sees the same switches: layout macros such as MADSIM_NH_PAIRS live in sim_kernel.h, which geometry.h shares)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -x hip -c madsim_hip.cpp -o /tmp/madsim_hip_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmadsim_hip_$tag.so /tmp/sim_kernel_$tag.o /tmp/madsim_hip_$tag.o
echo built ../libmadsim_hip_$tag.so
