#!/bin/bash
# round 6, experiment 2: the latency-first build (ONE wavefront per SIMD with the whole 512-register file: spills go to AGPRs) for
# tiny batches and the open batch behind single-patch callers, with and without proxies
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/e2; mkdir -p $O
for l in product lat0 lat2; do
  p=hpmvs_amd/libhpmvs_amd.so; [ $l != product ] && p=tools/ab/$l/libhpmvs_amd.so
  echo "== $l" | tee -a $O/single_call.txt
  timeout 300 python tools/single_call_latency.py $p 2>&1 | grep "^n " | tee -a $O/single_call.txt
done
CALLERS_THREADS=256,8,1 CALLERS_REPS=1 timeout 900 python tools/callers_ab.py lat0 lat1 lat2 2>&1 | tee $O/callers.txt | grep threads
