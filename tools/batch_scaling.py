"""Kernel time vs batch size on the C1 scene: separates the marginal per-patch cost from the ramp-down tail."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from hpmvs_amd import api, synth
dev = torch.device('cuda', 0)
scene = synth.make_scene(12, 1920, 1080, n_waves=24, device=dev)
g = api.Scene(scene)
seeds = synth.make_seeds(scene, 400000, start_level=4, max_images=32)
rows = []
for n in (12500, 25000, 50000, 100000, 200000, 400000):
    ms = []
    for _ in range(3):
        b = api.Batch.from_seeds(seeds, np.arange(n)); api.optimize_batch(g, b); ms.append(g.last_kernel_ms())
    rows.append((n, min(ms)))
    print(n, 'kernel ms', round(min(ms), 2), 'patches/s %.3g' % (n / min(ms) * 1e3), 'max evals', int(b.nevals.max()))
(n0, t0), (n1, t1) = rows[-2], rows[-1]
slope = (t1 - t0) / (n1 - n0)
print('marginal us/patch %.4f -> %.3g patches/s; intercept ms %.2f' % (slope * 1e3, 1e3 / slope, t1 - slope * n1))
