"""Launch time of the wide path (kernel_wide.hip) against the number of marked patches: 200-view scene at START_LEVEL 1."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from hpmvs_amd import api, synth
if len(sys.argv) > 1: api.LIB_PATH = sys.argv[1]
scene = synth.make_scene(200, 960, 540, n_waves=24, device=torch.device('cuda', 0))
g = api.Scene(scene)
seeds = synth.make_seeds(scene, 4096, start_level=1, max_images=256, seed=synth.SEED + 21)
for n in (1, 8, 64, 256, 1024, 2048, 4096):
    b = api.Batch.from_seeds(seeds, np.arange(n)); api.optimize_batch(g, b)
    ms = []
    for _ in range(2):
        b = api.Batch.from_seeds(seeds, np.arange(n)); api.optimize_batch(g, b); ms.append(g.last_kernel_ms())
    print('n', n, 'launch ms %.1f' % min(ms), 'wide', g.last_wide_patches(), 'ok', int(b.ok.sum()), 'evals max', int(b.nevals.max()), 'mean', float(b.nevals[b.nevals > 0].mean()),
          'grabs max', int(b.ngrabs.max()), 'mean', float(b.ngrabs.mean()), 'list max', int(b.n_images.max()), flush=True)
