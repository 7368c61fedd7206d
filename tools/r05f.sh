mkdir -p gpurun_out/r05f; O=gpurun_out/r05f
for lib in hpmvs_amd/libhpmvs_amd.so tools/ab/obj3/libhpmvs_amd.so tools/ab/obj4/libhpmvs_amd.so hpmvs_amd/libhpmvs_amd.so; do
  timeout 300 python tools/objective_time.py $lib 50 3840 2160 1000000 6 >> $O/obj_occupancy.txt 2>&1
done
grep -h "objective kernel" $O/obj_occupancy.txt
