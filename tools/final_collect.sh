#!/bin/bash
# final collection of round 4 (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04
timeout 1500 bash tools/collect_round.sh r04
O=gpurun_out/prof_r04
timeout 400 python bench.py --workload s1_50v_4k_1e6_level1 --steps 3 --warmup 1 --no-cpu > $O/bench_s1_level1.json 2>$O/bench_s1.err
for PMC in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d /tmp/s1_$PMC -o p -- python bench.py --workload s1_50v_4k_1e6_level1 --steps 1 --warmup 0 --no-cpu > $O/bench_s1_$PMC.log 2>&1
  python profiles/summarize.py /tmp/s1_$PMC $O/s1_level1_pmc_$PMC
done
timeout 300 python tools/single_call_latency.py > $O/single_call_latency_after.txt 2>&1
timeout 900 python tools/strong_scaling_projection.py 5 > $O/strong_scaling_projection.json 2>$O/proj.err
timeout 900 python tools/random_parity_sweep.py > $O/random_parity_sweep.txt 2>&1
tail -n 3 $O/random_parity_sweep.txt
tail -c 600 $O/strong_scaling_projection.json
