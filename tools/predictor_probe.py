"""Is the evaluation count predictable from quantities known before BOBYQA starts?"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from hpmvs_amd import api, synth
from oracle import oracle as orc
dev = torch.device('cuda', 0)
scene = synth.make_scene(12, 1920, 1080, n_waves=24, device=dev)
g = api.Scene(scene); osc = orc.OracleScene(scene)
n = 12000
seeds = synth.make_seeds(scene, n, start_level=4, max_images=32)
b = api.Batch.from_seeds(seeds); api.optimize_batch(g, b)
P = orc.patches_from_seeds(seeds)
x0 = np.array([orc.initial_parameters(osc, P[k]) for k in range(n)])
f0, _ = api.objective_batch(g, api.Batch.from_seeds(seeds), x0)
inc = api.inccs_batch(g, api.Batch.from_seeds(seeds), 0, 0)
ok = b.ok.astype(bool)
ne = b.nevals[ok].astype(float)
def rank(v): return np.argsort(np.argsort(v)).astype(float)
feats = {'f0': f0[ok], 'n_images_in': seeds.n_images[ok].astype(float), 'scale': seeds.scale[ok].astype(float),
         'mean_incc': np.array([inc[k, 1:seeds.n_images[k]].mean() for k in np.nonzero(ok)[0]]),
         'max_incc': np.array([inc[k, 1:seeds.n_images[k]].max() for k in np.nonzero(ok)[0]]),
         'n_final_images': b.n_images[ok].astype(float), 'fmin': b.fmin[ok]}
for k, v in feats.items():
    print('%-16s pearson %.3f spearman %.3f' % (k, np.corrcoef(v, ne)[0, 1], np.corrcoef(rank(v), rank(ne))[0, 1]))
print('nevals of successes: mean %.1f p50 %.0f p90 %.0f p99 %.0f' % (ne.mean(), *np.percentile(ne, [50, 90, 99])))
