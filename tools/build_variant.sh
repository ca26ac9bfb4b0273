#!/bin/bash
# Build the library with extra compiler flags as compatibility-family-learning_amd/lib/libcfl_hip_<name>.so, from an object directory
# of its own (the production objects and libcfl_hip.so are not touched).  Usage: tools/build_variant.sh <name> <flags...>
# e.g. tools/build_variant.sh nodp -DABL_NO_DP_PUSH ; tools/build_variant.sh xnosplit -DABL_X_NOSPLIT       (A/B: tools/ab_lib.sh)
set -e
name=$1; shift
cd "$(dirname "$0")/../compatibility-family-learning_amd"
mkdir -p build_$name
for n in cfl_hip cfl_conv cfl_gan cfl_eval cfl_dp; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC "$@" -c csrc/$n.hip -o build_$name/$n.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_$name/*.o -o lib/libcfl_hip_$name.so
echo lib/libcfl_hip_$name.so
