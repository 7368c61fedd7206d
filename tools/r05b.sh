mkdir -p gpurun_out/r05b; O=gpurun_out/r05b
timeout 1200 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest.log 2>&1; tail -12 $O/pytest.log
for lib in hpmvs_amd/libhpmvs_amd.so tools/ab/proxy0.so; do
  timeout 200 python tools/time_lib.py $lib >> $O/ab_c1.txt 2>&1
  timeout 300 python tools/time_lib.py $lib 50 3840 2160 1000000 4 >> $O/ab_c2.txt 2>&1
  timeout 300 python tools/time_lib.py $lib 50 3840 2160 125000 6 >> $O/ab_c2_125k.txt 2>&1
  timeout 200 python tools/time_lib.py $lib 12 1920 1080 4096 10 >> $O/ab_c1_4096.txt 2>&1
done
grep -h "kernel ms" $O/ab_*.txt
timeout 300 python tools/single_call_latency.py > $O/single_call_latency.txt 2>&1; cat $O/single_call_latency.txt
