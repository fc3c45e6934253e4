#!/bin/bash
# Diagnostic: builds libpqdet_hip_skipN.so variants whose fused-backward kernel leaves one phase out (XB_SKIP bit mask; results wrong on
# purpose) -- the time each phase costs inside pq_xdw_bwd's walking kernel.  Run on the build host, then scripts/irbwd_phases.py on the GPU.
set -e
cd "$(dirname "$0")/.."
L=pqdet_amd/lib
OBJS=$(ls $L/*.o | grep -v -e _stamps -e _skip -e irbwd.o)
for m in ${MASKS:-1 2 4 8 16 31}; do
  /opt/rocm/bin/hipcc -c pqdet_amd/csrc/irbwd.hip -o $L/irbwd_skip$m.o -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-result -Iinclude -DXB_SKIP=$m &
done
wait
for m in ${MASKS:-1 2 4 8 16 31}; do
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $L/libpqdet_hip_skip$m.so $OBJS $L/irbwd_skip$m.o
done
ls -la $L/*skip*.so
