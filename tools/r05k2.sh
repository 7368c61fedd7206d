#!/bin/bash
# pair loads (-DHPMVS_PAIR_LOADS=1: one 12-byte load per image row for two neighbouring texels) against the product, one box
mkdir -p gpurun_out/r05k2
{
for rep in 1 2; do
for v in base pair; do
  echo "== $v"
  timeout 300 python tools/objective_time.py tools/ab/$v/libhpmvs_amd.so 50 3840 2160 1000000 5 2>&1 | tail -1 | cut -c20-200
  timeout 400 python tools/time_lib.py tools/ab/$v/libhpmvs_amd.so 50 3840 2160 1000000 3 2>&1 | tail -1 | cut -c20-200
  timeout 200 python tools/time_lib.py tools/ab/$v/libhpmvs_amd.so 12 1920 1080 100000 6 2>&1 | tail -1 | cut -c20-200
done; done
} | tee gpurun_out/r05k2/ab.txt
