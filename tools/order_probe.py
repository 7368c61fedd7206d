import sys, numpy as np, torch
sys.path.insert(0, '.')
from hpmvs_amd import api, synth
dev = torch.device('cuda', 0)
scene = synth.make_scene(12, 1920, 1080, n_waves=24, device=dev)
g = api.Scene(scene)
seeds = synth.make_seeds(scene, 100000, start_level=4, max_images=32)
b = api.Batch.from_seeds(seeds); api.optimize_batch(g, b)
ne = b.nevals.astype(float)
print('nevals mean', ne.mean(), 'pct', np.percentile(ne,[50,90,99,99.9,100]))
for name, v in (('n_images_in', seeds.n_images.astype(float)), ('scale', seeds.scale.astype(float)), ('ok', b.ok.astype(float)), ('stage', b.stage.astype(float))):
    print(name, 'corr with nevals', np.corrcoef(v, ne)[0,1])
for k in range(2, 13):
    m = seeds.n_images == k
    if m.sum(): print('n_images', k, 'count', int(m.sum()), 'mean evals', ne[m].mean(), 'p99', np.percentile(ne[m], 99))
# what if long ones ran first: kernel time with an oracle ordering (descending nevals) vs ascending
for order_name, order in (('orig', np.arange(len(ne))), ('desc', np.argsort(-ne)), ('asc', np.argsort(ne))):
    bb = api.Batch.from_seeds(seeds, order); ms=[]
    for _ in range(3):
        bb = api.Batch.from_seeds(seeds, order); api.optimize_batch(g, bb); ms.append(g.last_kernel_ms())
    print(order_name, 'kernel ms', min(ms))
# realistic predictors: input n_images descending (stable), and random
rng = np.random.default_rng(0)
for order_name, order in (('nimg_desc', np.argsort(-seeds.n_images, kind='stable')), ('random', rng.permutation(len(ne))),
                          ('ok_first(oracle)', np.argsort(-b.ok.astype(int), kind='stable'))):
    ms=[]
    for _ in range(3):
        bb = api.Batch.from_seeds(seeds, order); api.optimize_batch(g, bb); ms.append(g.last_kernel_ms())
    print(order_name, 'kernel ms', min(ms))
