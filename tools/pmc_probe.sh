#!/bin/bash
# extra PMC passes (memory pipeline / LDS / instruction-level counters) of one bench.py launch; run ON THE GPU BOX:
#   tools/pmc_probe.sh <tag>   -> gpurun_out/probe_<tag>/*.json
set -u
TAG=${1:-x}
export TMPDIR=/tmp
OUT=gpurun_out/probe_$TAG; RAW=/tmp/rocprof_probe_$TAG
mkdir -p $OUT; rm -rf $RAW
i=0
for PMC in \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_IFETCH SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $RAW/pmc$i -o p -- python bench.py --steps 1 --warmup 0 --no-cpu > $OUT/bench_pmc$i.log 2>&1
  python profiles/summarize.py $RAW/pmc$i $OUT/probe_pmc$i
done
rm -rf $RAW
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/probe_*/probe_pmc*_counters.json')):
    d=json.load(open(f)).get('optimize_kernel',{})
    print(f.split('/')[-1], json.dumps(d.get('per_dispatch',{})))
P
