"""single-patch callers in a loop + one batch launch (see tests/test_gpu_optimize.py): how long does the batch take?
   mixed_use_check.py [threads] [copies]"""
import sys, time, threading, numpy as np
sys.path.insert(0, '.')
from hpmvs_amd import api, synth
T = int(sys.argv[1]) if len(sys.argv) > 1 else 16
COP = int(sys.argv[2]) if len(sys.argv) > 2 else 100
scene = synth.make_scene(3, 640, 480, n_waves=24)
seeds = synth.make_seeds(scene, 200, start_level=2)
g = api.Scene(scene)
n = len(seeds.scale)
ref = api.Batch.from_seeds(seeds, np.arange(n)); api.optimize_batch(g, ref)
stop = threading.Event(); calls = [0] * T; lat = [[] for _ in range(T)]
def caller(k):
    i = 0
    while not stop.is_set():
        b = api.Batch.from_seeds(seeds, np.array([(k * 7 + i) % n]))
        t0 = time.time(); api.optimize_batch(g, b); lat[k].append(time.time() - t0)
        i += 1; calls[k] = i
th = [threading.Thread(target=caller, args=(k,)) for k in range(T)]
for t in th: t.start()
time.sleep(0.5)
print('calls before', sum(calls), flush=True)
for rep in range(3):
    big = api.Batch.from_seeds(seeds, np.tile(np.arange(n), COP))
    t0 = time.time(); api.optimize_batch(g, big); dt = time.time() - t0
    print('batch of', big.n, 'took %.3f s' % dt, 'kernel ms', g.last_kernel_ms(), 'calls so far', sum(calls), flush=True)
    time.sleep(0.3)
stop.set()
for t in th: t.join(timeout=60)
allat = np.concatenate([np.array(l) for l in lat])
print('single-call latency ms: median %.2f p99 %.2f max %.2f' % (np.median(allat) * 1e3, np.percentile(allat, 99) * 1e3, allat.max() * 1e3))
