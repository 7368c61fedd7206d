#!/bin/bash
# compiler scheduling strategies for the whole library (-mllvm ... / -O2), one box: C2 (3 launches) and C1 (6 launches) each
mkdir -p gpurun_out/r05j
{
for v in base O2 Os O2nounroll O3nounroll O1 base; do
  [ -f tools/ab/$v/libhpmvs_amd.so ] || continue
  echo "== $v"
  timeout 400 python tools/time_lib.py tools/ab/$v/libhpmvs_amd.so 50 3840 2160 1000000 3 2>&1 | tail -1 | cut -c30-200
  timeout 200 python tools/time_lib.py tools/ab/$v/libhpmvs_amd.so 12 1920 1080 100000 6 2>&1 | tail -1 | cut -c30-200
done
} | tee gpurun_out/r05j/ab.txt
