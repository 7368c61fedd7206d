#!/bin/bash
# PMC passes over the two kernels behind hpmvs_objective_batch (tools/objective_ab.py runs both on the same requests): what the
# vector-memory path (address unit TA, vector L1 TCP) does while the 8-byte tap loads of the windows go through it.
#   bash tools/pmc_objective_mem.sh <out_dir>      (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
export O=${1:-gpurun_out/pmc_objective_mem}; mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -o -E "\b(TA|TCP|TD|SQ|TCC)_[A-Za-z0-9_]+" | sort -u > $O/counters_available.txt
i=0
# (at most two counters of a block per pass: "Request exceeds the capabilities of the hardware" otherwise, and rocprofv3 then hangs
#  until the timeout)
for PMC in "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TAGRAM0_REQ_sum TCP_TOTAL_ACCESSES_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d /tmp/pmcmem$i -o p -- python tools/objective_ab.py 50 3840 2160 1000000 2 > $O/run$i.log 2>&1
  python profiles/summarize.py /tmp/pmcmem$i $O/mem_pmc$i objective_lane_kernel objective_kernel
done
python - <<'PY'
import json, glob, os
O = os.environ["O"]
c = {}
for f in sorted(glob.glob(O + "/mem_pmc*_counters.json")):
    for k, v in json.load(open(f)).items():
        c.setdefault(k, {}).update(v.get("per_dispatch", {}))
json.dump(c, open(O + "/objective_mem_counters.json", "w"), indent=1, sort_keys=True)
for k in c:
    print(k)
    for n in sorted(c[k]): print(f"   {n:44s} {c[k][n]:18.0f}")
PY
