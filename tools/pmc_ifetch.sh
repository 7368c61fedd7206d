#!/bin/bash
# Instruction fetch of the refinement kernel (its code is ~180 KB: bq_step_ 87 KB, the kernel body 44 KB, trsbox 22 KB, ... against a
# 64 KB instruction cache shared by two CUs):  bash tools/pmc_ifetch.sh <out_dir>
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
export O=${1:-gpurun_out/pmc_ifetch}; mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -o -E "\b[A-Z][A-Za-z0-9]*_[A-Za-z0-9_]+" | sort -u > $O/all_counter_names.txt
i=0
for PMC in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQC_TC_INST_REQ SQC_ICACHE_INPUT_VALID_READYB SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INFLIGHT_LEVEL"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d /tmp/pmcif$i -o p -- python bench.py --steps 1 --warmup 0 --no-cpu > $O/run$i.log 2>&1
  python profiles/summarize.py /tmp/pmcif$i $O/if_pmc$i optimize_kernel
  grep -v "^    @" $O/run$i.log | grep -E "error code|Could not" | head -3
done
python - <<'PY'
import json, glob, os
O = os.environ["O"]
c = {}
for f in sorted(glob.glob(O + "/if_pmc*_counters.json")):
    for k, v in json.load(open(f)).items():
        c.update(v.get("per_dispatch", {}))
json.dump(c, open(O + "/ifetch_counters.json", "w"), indent=1, sort_keys=True)
for n in sorted(c): print(f"   {n:44s} {c[n]:18.0f}")
PY
grep -i -E "icache|ifetch|SQC_" $O/all_counter_names.txt | tr '\n' ' '
