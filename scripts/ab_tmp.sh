#!/bin/bash
# temporary A/B driver: start stagger of the fused narrow backward
mkdir -p gpurun_out/r4l
for pct in 0 50 100 150; do
  echo "== PQ_PW_STAGGER=$pct"
  PQ_PW_STAGGER=$pct PQDET_HIP_LIB=pqdet_amd/lib/libpqdet_hip_stamps.so timeout -k 10 300 python scripts/pw_stamps.py 2>&1 | grep "^M=" | cut -c1-330
done
