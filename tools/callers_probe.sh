#!/bin/bash
python - <<'PY'
import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from hpmvs_amd import synth
import test_gpu_cpp_interface as t
scene = synth.make_scene(3, 640, 480, n_waves=16)
n=10000
seeds = synth.make_seeds(scene, n, start_level=2, seed=synth.SEED + 3)
t._dump_scene('/tmp/scene10k.bin', scene, seeds, n)
PY
for T in 1 16 64 256; do echo "== T=$T"; HPMVS_SERVICE=1 HPMVS_SERVICE_STATS=1 HPMVS_SERVICE_LANES=1 timeout 300 tests/native/bench_callers /tmp/scene10k.bin $T 2>&1 | grep -E "hpmvs service|patches" | cut -c1-330; done
