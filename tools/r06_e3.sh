#!/bin/bash
# round 6, experiment 3: does the order of the batch matter?  bench.py --order tiles = host-side bucket sort of the round by
# (reference image, 256-px tile) against the seeds' own (Halton) order
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/e3; mkdir -p $O
for w in c2_50v_4k_1e6 s1_50v_4k_1e6_level1 c1_12v_1080p_1e5; do
  for o in as-is tiles; do
    timeout 600 python bench.py --workload $w --order $o --steps 5 --warmup 2 --no-cpu > $O/${w}_$o.json 2>$O/${w}_$o.err
    python - <<PY
import json
d=json.loads(open("$O/${w}_$o.json").read().strip().splitlines()[-1])
print("$w", "$o", "kernel ms %.2f"%d["roofline"]["kernel_ms"], "value %.3f M"%(d["value"]/1e6), "ok %.4f"%d["success_fraction"])
PY
  done
done
