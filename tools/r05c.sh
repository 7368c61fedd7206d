mkdir -p gpurun_out/r05c; O=gpurun_out/r05c
timeout 900 python tools/callers_ab.py p0 > $O/callers_ab.txt 2>&1; cat $O/callers_ab.txt
for lib in hpmvs_amd/libhpmvs_amd.so tools/ab/p0/libhpmvs_amd.so; do
  timeout 200 python tools/time_lib.py $lib >> $O/ab_c1.txt 2>&1
  timeout 300 python tools/time_lib.py $lib 50 3840 2160 1000000 4 >> $O/ab_c2.txt 2>&1
done
grep -h "kernel ms" $O/ab_*.txt
