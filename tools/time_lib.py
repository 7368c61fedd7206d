"""kernel time of one launch for an alternative build of the library:
   time_lib.py <lib.so> [views w h n reps]   (default: the C1 scene, 12 x 1080p, 1e5 seeds, 10 repetitions)"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from hpmvs_amd import api, synth
api.LIB_PATH = sys.argv[1]
V, W_, H_, N_, R_ = (int(a) for a in sys.argv[2:7]) if len(sys.argv) > 6 else (12, 1920, 1080, 100000, 10)
dev = torch.device('cuda', 0)
scene = synth.make_scene(V, W_, H_, n_waves=24, device=dev)
g = api.Scene(scene)
seeds = synth.make_seeds(scene, N_, start_level=4, max_images=32)
ms = []
for _ in range(R_):
    b = api.Batch.from_seeds(seeds); api.optimize_batch(g, b); ms.append(g.last_kernel_ms())
print(sys.argv[1], 'kernel ms min %.2f median %.2f' % (min(ms), float(np.median(ms))), [round(m, 1) for m in ms], 'ok', b.ok.mean(), 'checksum', float(b.center[b.ok.astype(bool)].sum()))
