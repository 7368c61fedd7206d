#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + separate PMC passes of bench.py,
# reduced to small summaries under gpurun_out/prof_<tag>/ (copy the ones to be judged into profiles/).
#   profiles/collect.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
ARGS=${@:---steps 3 --warmup 1 --no-cpu}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG; RAW=/tmp/rocprof_raw_$TAG
mkdir -p $OUT; rm -rf $RAW
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o t -- python bench.py $ARGS > $OUT/bench_trace.log 2>&1
python profiles/summarize.py $RAW/trace $OUT/${TAG}_trace
find $RAW/trace -name '*kernel_stats.csv' | head -1 | xargs -r -I{} sh -c 'head -12 {} > '$OUT/${TAG}'_rocprof_kernel_stats_top.csv'
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $RAW/pmc$i -o p -- python bench.py --steps 1 --warmup 0 --no-cpu ${BENCH_EXTRA:-} > $OUT/bench_pmc$i.log 2>&1
  python profiles/summarize.py $RAW/pmc$i $OUT/${TAG}_pmc$i
done
rm -rf $RAW
ls -la $OUT
