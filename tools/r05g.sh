mkdir -p gpurun_out/r05g; O=gpurun_out/r05g
for lib in hpmvs_amd/libhpmvs_amd.so tools/ab/js1/libhpmvs_amd.so tools/ab/js3/libhpmvs_amd.so hpmvs_amd/libhpmvs_amd.so; do
  timeout 200 python tools/time_lib.py $lib >> $O/ab_c1.txt 2>&1
  timeout 300 python tools/time_lib.py $lib 50 3840 2160 1000000 4 >> $O/ab_c2.txt 2>&1
done
grep -h "kernel ms\|rror" $O/ab_*.txt
