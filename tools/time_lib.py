import sys, numpy as np, torch
sys.path.insert(0, '.')
from hpmvs_amd import api, synth
api.LIB_PATH = sys.argv[1]
dev = torch.device('cuda', 0)
scene = synth.make_scene(12, 1920, 1080, n_waves=24, device=dev)
g = api.Scene(scene)
seeds = synth.make_seeds(scene, 100000, start_level=4, max_images=32)
ms = []
for _ in range(10):
    b = api.Batch.from_seeds(seeds); api.optimize_batch(g, b); ms.append(g.last_kernel_ms())
print(sys.argv[1], 'kernel ms min %.2f median %.2f' % (min(ms), float(np.median(ms))), [round(m, 1) for m in ms], 'ok', b.ok.mean(), 'checksum', float(b.center[b.ok.astype(bool)].sum()))
