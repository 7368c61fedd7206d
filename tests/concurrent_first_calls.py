"""Run in a FRESH process by tests/test_gpu_concurrent_first_calls.py: T host threads make their first single-patch
hpmvs_optimize_batch calls at the same time -- the moment at which the scene grows its second, third ... launch workspace -- on a
dense scene whose patches go through the wide kernel, and every call must return the patch that one batch over all of them returns.
(Round 6: the fill of a newly grown workspace was issued on the null stream, which is not ordered with the non-blocking streams small
host batches are launched on; it ran into the first launch that used the workspace and zeroed its image-list rows -- one first call
in a few came back with another refinement.)   usage: concurrent_first_calls.py [rows threads reps]"""
import sys, threading, numpy as np, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hpmvs_amd import api, synth
if os.environ.get('HPMVS_TEST_LIB'):   # (a variant build: showing that the test fails without the fix)
    api.LIB_PATH = os.environ['HPMVS_TEST_LIB']
import torch
scene = synth.make_scene(120, 640, 360, n_waves=16, device=torch.device('cuda', 0))   # (rendered on the GPU: seconds instead of half a minute)
n = 160
seeds = synth.make_seeds(scene, n, start_level=1, max_images=256, seed=synth.SEED + 22)
g = api.Scene(scene)
ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 120
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 6
def run(idx, rows):
    b = api.Batch.from_seeds(seeds, np.array(idx))
    b = api.Batch(b.center, b.normal, b.scale, b.n_images, b.images[:, :rows])
    api.optimize_batch(g, b)
    return b
full = run(list(range(n)), 120)
print('batch: ok', int(full.ok.sum()), 'evals', int(full.nevals.sum()))
bad = []
lock = threading.Lock()
def worker(t):
    for rep in range(REPS):
        for k in range(t, 40, T):
            one = run([k], ROWS)
            fits = full.n_images[k] <= ROWS
            if fits:
                same = one.ok[0] == full.ok[k] and np.array_equal(one.center[0], full.center[k]) and one.nevals[0] == full.nevals[k]
            else:
                same = one.ok[0] == 0 and one.stage[0] == 100
            if not same:
                with lock: bad.append((k, rep, int(one.ok[0]), int(one.stage[0]), int(one.n_images[0]), int(one.nevals[0]), int(full.n_images[k]), int(full.nevals[k])))
ths = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
[t.start() for t in ths]; [t.join() for t in ths]
print('rows', ROWS, 'threads', T, 'calls', 40 * REPS, 'mismatches', len(bad), bad[:12])
sys.exit(1 if bad else 0)
