#!/bin/bash
# round 6, experiment 1: what phase A costs under the register budget and wave count an optimiser-only kernel would have
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/e1; mkdir -p $O
for l in base w4r2 w4r3; do timeout 300 python tools/time_lib.py tools/ab/e1/$l.so 50 3840 2160 1000000 3 > $O/time_$l.txt 2>&1; tail -1 $O/time_$l.txt; done
for l in prof_base prof_w4r2 prof_w4r3; do timeout 300 python tools/phase_profile.py tools/ab/e1/$l.so 50 3840 2160 1000000 > $O/$l.txt 2>&1; grep -E "bq_step|kernel ms|A share|region|list services" $O/$l.txt; done
