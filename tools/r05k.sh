mkdir -p gpurun_out/r05k; O=gpurun_out/r05k
python - > $O/callers_spin.txt 2>&1 <<'PY'
import json, os, subprocess, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from hpmvs_amd import synth
from test_gpu_cpp_interface import _dump_scene
scene = synth.make_scene(3, 640, 480, n_waves=24)
n = 10000
seeds = synth.make_seeds(scene, n, start_level=2, seed=synth.SEED + 3)
dump = "/tmp/scene10k.bin"
_dump_scene(dump, scene, seeds, n)
exe = "tests/native/bench_callers"
for spin in ("a", "b"):
    for threads in (512, 256, 64, 8):
        env = dict(os.environ, HPMVS_SERVICE="1", OMP_WAIT_POLICY="passive", HPMVS_SERVICE_STATS="1", HPMVS_X=spin)
        r = subprocess.run([exe, dump, str(threads)], capture_output=True, text=True, timeout=900, env=env)
        if r.returncode != 0:
            print("spin", spin, threads, "FAILED", r.stdout[-300:], r.stderr[-300:]); continue
        row = json.loads(r.stdout.strip().splitlines()[-1])
        print("spin", spin, "threads", threads, "calls/s %.0f" % row["calls_patches_per_s"], "loop s %.3f cpu user %.3f sys %.3f" % (row["calls_s"], row["loop_cpu_user_s"], row["loop_cpu_sys_s"]), "|", r.stderr.strip().splitlines()[-1][40:230], flush=True)
PY
cat $O/callers_spin.txt
