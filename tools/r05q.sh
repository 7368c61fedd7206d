mkdir -p gpurun_out/r05q; O=gpurun_out/r05q; rm -f $O/*.txt
for lib in hpmvs_amd/libhpmvs_amd.so tools/ab/sf1/libhpmvs_amd.so tools/ab/sp1/libhpmvs_amd.so tools/ab/sfsp/libhpmvs_amd.so hpmvs_amd/libhpmvs_amd.so; do
  timeout 300 python tools/time_lib.py $lib 50 3840 2160 1000000 3 >> $O/ab_c2.txt 2>&1
  HPMVS_SLOTS=64 timeout 300 python tools/time_lib.py $lib 50 3840 2160 250000 4 >> $O/ab_250k.txt 2>&1
done
grep -h "kernel ms\|rror" $O/ab_c2.txt $O/ab_250k.txt | cut -c1-150
