#!/bin/bash
# A round's last collection (run on the GPU box from the repo root): collect_round.sh, then the START_LEVEL-1 stress line
# with its two PMC passes, the C1 PMC passes, single-call latency, the strong-scaling projection and the random parity sweep.
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/prof_$TAG
timeout 1700 bash tools/collect_round.sh $TAG || echo "collect_round.sh reported failures"
ok() { ! grep -q "^Traceback" "$1" 2>/dev/null; }
timeout 400 python bench.py --workload s1_50v_4k_1e6_level1 --steps 3 --warmup 1 --no-cpu > $O/bench_s1_level1.json 2>$O/bench_s1.err || rm -f $O/bench_s1_level1.json
for PMC in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d /tmp/s1_$PMC -o p -- python bench.py --workload s1_50v_4k_1e6_level1 --steps 1 --warmup 0 --no-cpu > $O/bench_s1_$PMC.log 2>&1
  python profiles/summarize.py /tmp/s1_$PMC $O/s1_level1_pmc_$PMC
done
# the C1 record of profiles/pmc_traffic.json from THIS build (the C1 record was a round old)
mkdir -p gpurun_out/prof_${TAG}c1
BENCH_EXTRA="--workload c1_12v_1080p_1e5" timeout 900 bash profiles/collect.sh ${TAG}c1 --workload c1_12v_1080p_1e5 --steps 3 --warmup 1 --no-cpu > gpurun_out/collect_${TAG}c1.log 2>&1
timeout 60 python profiles/make_pmc_record.py gpurun_out/prof_${TAG}c1 ${TAG}c1 $O/pmc_traffic.json | tail -2
for f in single_call_latency:"python tools/single_call_latency.py" strong_scaling_projection:"python tools/strong_scaling_projection.py 5" random_parity_sweep:"python tools/random_parity_sweep.py"; do
  name=${f%%:*}; cmd=${f#*:}; ext=txt; [ $name = strong_scaling_projection ] && ext=json
  timeout 900 $cmd > $O/$name.$ext 2>$O/$name.err
  ok $O/$name.$ext && ok $O/$name.err || { echo "$name FAILED"; mv $O/$name.$ext $O/$name.FAILED; }
done
tail -n 3 $O/random_parity_sweep.txt
tail -c 600 $O/strong_scaling_projection.json
