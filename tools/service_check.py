import sys, numpy as np
sys.path.insert(0,'.')
from hpmvs_amd import api, synth
scene = synth.make_scene(3, 640, 480, n_waves=16)
seeds = synth.make_seeds(scene, 64, start_level=2, seed=synth.SEED + 3)
g = api.Scene(scene, device=0)
full = api.Batch.from_seeds(seeds); api.optimize_batch(g, full)   # n = 64 -> staged path
bad = 0
for k in range(64):
    one = api.Batch.from_seeds(seeds, np.array([k])); api.optimize_batch(g, one)   # n = 1 -> open batch
    for f in ("ok","center","normal","n_images","images","color","ncc","fmin","x","result","nevals","stage","ngrabs"):
        a, b = getattr(one, f)[0], getattr(full, f)[k]
        if not np.array_equal(a, b):
            if bad < 12: print(k, f, a, b)
            bad += 1
print('mismatching fields', bad)
