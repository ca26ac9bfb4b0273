#!/bin/bash
# Instrumented build for tools/stamp_probe.py / mid_stamp_probe.py: the library with -DCFL_STAMPS (per-wave cycle-counter stamps in
# the proj / mid kernels) as compatibility-family-learning_amd/lib/stamps.so, from its own object directory -- the production
# objects and libcfl_hip.so are not touched.  Run here (hipcc cross-compiles), the .so travels with gpurun.
set -e
cd "$(dirname "$0")/../compatibility-family-learning_amd"
mkdir -p build_stamps
for n in cfl_hip cfl_conv cfl_gan cfl_eval cfl_dp; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -DCFL_STAMPS -c csrc/$n.hip -o build_stamps/$n.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_stamps/*.o -o lib/stamps.so
echo lib/stamps.so
