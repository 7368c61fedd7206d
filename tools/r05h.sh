#!/bin/bash
# A/B in one call: product build against -DHPMVS_SHARED_SUB=1 (one subtraction per texel value for variance and normalisation)
mkdir -p gpurun_out/r05h
for rep in 1 2; do
for v in base sub; do
  timeout 300 python tools/time_lib.py tools/ab/$v/libhpmvs_amd.so 2>&1 | tail -1
  timeout 600 python tools/time_lib.py tools/ab/$v/libhpmvs_amd.so 50 3840 2160 1000000 4 2>&1 | tail -1
done; done | tee gpurun_out/r05h/ab.txt
for v in base sub; do timeout 300 python tools/objective_time.py tools/ab/$v/libhpmvs_amd.so 50 3840 2160 1000000 5 2>&1 | tail -1; done | tee -a gpurun_out/r05h/ab.txt
