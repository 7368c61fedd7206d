#!/bin/bash
# PMC passes over the stand-alone objective kernel (= the photometric service alone): where its wave-cycles go.
#   bash tools/pmc_objective.sh <out_dir>      (run on the GPU box from the repo root)
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=${1:-gpurun_out/pmc_objective}; mkdir -p $O
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F32"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d /tmp/pmcobj$i -o p -- python tools/objective_time.py hpmvs_amd/libhpmvs_amd.so 50 3840 2160 1000000 3 > $O/run$i.log 2>&1
  python profiles/summarize.py /tmp/pmcobj$i $O/obj_pmc$i objective_kernel
done
python - <<'PY'
import json, glob, os
O = os.environ.get("O", "gpurun_out/pmc_objective")
c = {}
for f in sorted(glob.glob(O + "/obj_pmc*_counters.json")):
    c.update(json.load(open(f)).get("objective_kernel", {}).get("per_dispatch", {}))
json.dump(c, open(O + "/objective_counters.json", "w"), indent=1, sort_keys=True)
wc = c.get("SQ_WAVE_CYCLES", 1.0)
for k in sorted(c):
    print(f"{k:34s} {c[k]:16.0f}  per wave-cycle {c[k] / wc:8.4f}")
PY
