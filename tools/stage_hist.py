import sys, numpy as np, torch
sys.path.insert(0, '.')
from hpmvs_amd import api, synth
dev = torch.device('cuda', 0)
for V, n in ((200, 200000), (50, 200000)):
    scene = synth.make_scene(V, 3840, 2160, n_waves=24, device=dev)
    g = api.Scene(scene)
    seeds = synth.make_seeds(scene, n, start_level=4, max_images=32)
    b = api.Batch.from_seeds(seeds); api.optimize_batch(g, b)
    st, cnt = np.unique(b.stage, return_counts=True)
    print(V, 'views: ok', b.ok.mean(), 'stage hist', dict(zip(st.tolist(), (cnt / n).round(4).tolist())), 'in n_images mean', seeds.n_images.mean(), 'out n_images mean', b.n_images[b.ok.astype(bool)].mean(), 'max', b.n_images[b.ok.astype(bool)].max(), 'kernel ms', g.last_kernel_ms())
    g.close()
