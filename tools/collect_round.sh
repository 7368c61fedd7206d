#!/bin/bash
# Everything profiles/<round>/ holds, in one gpurun call (run ON THE GPU BOX, from the repo root, under `timeout`):
#   timeout 1500 bash tools/collect_round.sh r03
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/prof_$TAG
timeout 1000 bash profiles/collect.sh $TAG --steps 3 --warmup 1 --no-cpu > gpurun_out/collect_$TAG.log 2>&1
timeout 60 python profiles/make_pmc_record.py $O $TAG $O/pmc_traffic.json | tail -3
timeout 200 python tools/phase_profile.py tools/libhpmvs_prof.so 50 3840 2160 1000000 > $O/phase_profile_c2.txt 2>&1
timeout 200 python tools/phase_profile.py tools/libhpmvs_prof.so > $O/phase_profile_c1.txt 2>&1
timeout 300 python bench.py --workload c1_12v_1080p_1e5 --steps 5 --warmup 1 > $O/bench_c1.json 2>$O/bench_c1.err
timeout 400 python bench.py --steps 5 --warmup 1 > $O/bench_c2.json 2>$O/bench_c2.err
timeout 600 python bench.py --workload c4_200v_4k_1e7 --steps 2 --warmup 1 --cpu-sample 16384 --cpu-timing-sample 131072 > $O/bench_c4_n1.json 2>$O/bench_c4_n1.err
tail -c 400 $O/bench_c2.json; echo; tail -c 300 $O/bench_c4_n1.json
