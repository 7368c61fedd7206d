mkdir -p gpurun_out/r05d; O=gpurun_out/r05d
timeout 900 python -m pytest tests/test_gpu_photometric.py tests/test_gpu_optimize.py tests/test_gpu_big_configs.py -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for lib in hpmvs_amd/libhpmvs_amd.so tools/ab/scan0/libhpmvs_amd.so; do
  timeout 200 python tools/time_lib.py $lib >> $O/ab_c1.txt 2>&1
  timeout 300 python tools/time_lib.py $lib 50 3840 2160 1000000 4 >> $O/ab_c2.txt 2>&1
  timeout 300 python tools/objective_time.py $lib 50 3840 2160 1000000 5 >> $O/ab_obj.txt 2>&1
done
for lib in hpmvs_amd/libhpmvs_amd.so tools/ab/scan0/libhpmvs_amd.so; do
  timeout 300 python tools/time_lib.py $lib 50 3840 2160 1000000 4 >> $O/ab_c2.txt 2>&1
done
grep -h "kernel ms" $O/ab_*.txt
