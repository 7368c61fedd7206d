mkdir -p gpurun_out/r05e; O=gpurun_out/r05e
for lib in hpmvs_amd/libhpmvs_amd.so tools/ab/w3/libhpmvs_amd.so; do
  timeout 200 python tools/time_lib.py $lib >> $O/ab_c1.txt 2>&1
  timeout 300 python tools/time_lib.py $lib 50 3840 2160 1000000 4 >> $O/ab_c2.txt 2>&1
  timeout 300 python tools/time_lib.py $lib 50 3840 2160 125000 5 >> $O/ab_c2_125k.txt 2>&1
done
grep -h "kernel ms\|rror" $O/ab_*.txt
timeout 300 python tools/phase_profile.py tools/libhpmvs_prof.so 50 3840 2160 1000000 > $O/phase_profile_c2.txt 2>&1; tail -45 $O/phase_profile_c2.txt
