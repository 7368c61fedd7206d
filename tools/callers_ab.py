"""Unchanged single-patch callers (tests/native/bench_callers) with the product library and with variant builds placed
under tools/ab/<name>/libhpmvs_amd.so (LD_LIBRARY_PATH comes before the binary's RUNPATH): callers_ab.py [names...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from hpmvs_amd import synth
from test_gpu_cpp_interface import _dump_scene
scene = synth.make_scene(3, 640, 480, n_waves=24)
n = 10000
seeds = synth.make_seeds(scene, n, start_level=2, seed=synth.SEED + 3)
dump = "/tmp/scene10k.bin"
_dump_scene(dump, scene, seeds, n)
exe = os.path.join(ROOT, "tests", "native", "bench_callers")
for name in ["product"] + sys.argv[1:]:
    env = dict(os.environ, HPMVS_SERVICE="1")
    if name != "product":
        env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "tools", "ab", name) + ":" + env.get("LD_LIBRARY_PATH", "")
    for rep in range(int(os.environ.get('CALLERS_REPS', '2'))):
        for threads in [int(t) for t in os.environ.get('CALLERS_THREADS', '512,256,64,8,1').split(',')]:
            r = subprocess.run([exe, dump, str(threads)], capture_output=True, text=True, timeout=900, env=env)
            if r.returncode != 0:
                print(name, threads, "FAILED", r.stderr[-300:]); continue
            row = json.loads(r.stdout.strip().splitlines()[-1])
            print(name, "threads", threads, "calls/s %.0f" % row["calls_patches_per_s"], "mean call ms %.3f" % row["single_call_ms_mean"],
                  "mismatches", row["mismatches"], flush=True)
