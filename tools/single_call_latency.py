"""Wall time of the host-pointer C ABI call for tiny batches (the unchanged-caller form: optimize() = batch of 1)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from hpmvs_amd import api, synth
if len(sys.argv) > 1: api.LIB_PATH = sys.argv[1]   # a variant build (tools/ab/<name>/libhpmvs_amd.so)
dev = torch.device('cuda', 0)
scene = synth.make_scene(12, 1920, 1080, n_waves=24, device=dev)
g = api.Scene(scene)
seeds = synth.make_seeds(scene, 4096, start_level=4, max_images=32)
for n in (1, 8, 64, 512, 4096):
    ts, ks = [], []
    for rep in range(6):
        b = api.Batch.from_seeds(seeds, np.arange(rep * n % 2048, rep * n % 2048 + n))
        t0 = time.perf_counter(); api.optimize_batch(g, b); ts.append(time.perf_counter() - t0)
        try:
            ks.append(g.last_kernel_ms())
        except api.HpmvsError:  # batches of <= 4 go through the open batch: no launch of their own
            ks.append(float('nan'))
    print('n', n, 'wall ms min %.2f' % (min(ts) * 1e3), 'kernel ms min %.2f' % min(ks), 'evals', int(b.nevals.max()))
