#!/bin/bash
# balanced quota (a workgroup's share split evenly between its wavefronts when it does not fill them) in the 64-slot build,
# against the product's choice (29 slots below 768 patches per CU), one box
mkdir -p gpurun_out/r05i
{
for n in 60000 100000 125000 150000; do
  for sc in "12 1920 1080" "50 3840 2160"; do
    echo "n $n scene $sc: product | balanced, 64 slots forced | product, 64 slots forced | balanced (own choice)"
    timeout 300 python tools/time_lib.py tools/ab/base/libhpmvs_amd.so $sc $n 5 2>&1 | tail -1 | cut -c30-120
    HPMVS_SLOTS=64 timeout 300 python tools/time_lib.py tools/ab/balq/libhpmvs_amd.so $sc $n 5 2>&1 | tail -1 | cut -c30-120
    HPMVS_SLOTS=64 timeout 300 python tools/time_lib.py tools/ab/base/libhpmvs_amd.so $sc $n 5 2>&1 | tail -1 | cut -c30-120
    timeout 300 python tools/time_lib.py tools/ab/balq/libhpmvs_amd.so $sc $n 5 2>&1 | tail -1 | cut -c30-120
  done
done
} | tee gpurun_out/r05i/ab.txt
