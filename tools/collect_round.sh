#!/bin/bash
# Everything profiles/<round>/ holds, in one gpurun call (run ON THE GPU BOX, from the repo root, under `timeout`):
#   timeout 1500 bash tools/collect_round.sh r05
# Before the call, HERE: `make -C hpmvs_amd/csrc && make -C hpmvs_amd/csrc prof` (the profiling library travels prebuilt).
# Every file is produced through `keep`, which refuses a failed command's output (a Python traceback is not a profile)
# and stamps the file with hpmvs_build_id() of the library that produced it.
set -u
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/prof_$TAG
mkdir -p $O
FAILED=0
ID=$(python -c "from hpmvs_amd import api; print(api.build_id())") || { echo "cannot load the product library"; exit 1; }
PID=$(python -c "
import ctypes as C
L = C.CDLL('tools/libhpmvs_prof.so'); L.hpmvs_build_id.restype = C.c_char_p
print(L.hpmvs_build_id().decode())") || { echo "tools/libhpmvs_prof.so does not load: run 'make -C hpmvs_amd/csrc prof' before gpurun"; exit 1; }
if [ "$ID" != "$PID" ]; then echo "tools/libhpmvs_prof.so is build $PID, the product library is $ID: rebuild it (make -C hpmvs_amd/csrc prof)"; exit 1; fi
echo "library build $ID" | tee $O/BUILD_ID.txt

# keep <seconds> <file> <command...>: stdout -> file only if the command succeeded and did not print a traceback
keep() {
  local secs=$1 dst=$2; shift 2
  local tmp=$(mktemp)
  timeout $secs "$@" > $tmp 2> $tmp.err
  local status=$?   # (the command's own status, not the grep's: ADVICE r5)
  if [ $status = 0 ] && ! grep -q "^Traceback" $tmp $tmp.err; then
    { cat $tmp; } > $dst
    case $dst in *.json) ;; *) echo "# library build $ID" >> $dst ;; esac
  else
    echo "FAILED ($status): $*" | tee -a $O/FAILED.txt; tail -5 $tmp.err | tee -a $O/FAILED.txt
    FAILED=1; rm -f $dst
  fi
  rm -f $tmp $tmp.err
}

timeout 1000 bash profiles/collect.sh $TAG --steps 3 --warmup 1 --no-cpu > gpurun_out/collect_$TAG.log 2>&1 || { echo "profiles/collect.sh failed" | tee -a $O/FAILED.txt; FAILED=1; }
timeout 60 python profiles/make_pmc_record.py $O $TAG $O/pmc_traffic.json | tail -3 || { echo "make_pmc_record failed" | tee -a $O/FAILED.txt; FAILED=1; }
keep 200 $O/phase_profile_c2.txt python tools/phase_profile.py tools/libhpmvs_prof.so 50 3840 2160 1000000
keep 200 $O/phase_profile_c1.txt python tools/phase_profile.py tools/libhpmvs_prof.so
keep 300 $O/bench_c1.json python bench.py --workload c1_12v_1080p_1e5 --steps 5 --warmup 1
keep 400 $O/bench_c2.json python bench.py --steps 5 --warmup 1
keep 600 $O/bench_c4_n1.json python bench.py --workload c4_200v_4k_1e7 --steps 2 --warmup 1 --cpu-sample 16384 --cpu-timing-sample 131072
[ -f $O/bench_c2.json ] && tail -c 400 $O/bench_c2.json; echo
[ $FAILED = 0 ] && echo "collect_round: all files written" || { echo "collect_round: SOME FILES ARE MISSING (see $O/FAILED.txt)"; exit 1; }
