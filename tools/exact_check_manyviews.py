"""Bit-exactness of the multi-chunk evaluation path: a 24-view scene where patches attach 9-24 images."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from hpmvs_amd import api, synth
from oracle import oracle as orc
dev = torch.device('cuda', 0)
scene = synth.make_scene(24, 960, 540, n_waves=24, device=dev)
g = api.Scene(scene); osc = orc.OracleScene(scene)
for sl in (1, 2):
    seeds = synth.make_seeds(scene, 400, start_level=sl, max_images=64, max_seed_images=24, seed=synth.SEED + sl)
    idx = np.nonzero(seeds.n_images >= 3)[0]
    print('level', sl, 'n_images histogram (seed lists):', np.bincount(seeds.n_images[idx])[-8:], 'max', seeds.n_images.max())
    batch = api.Batch.from_seeds(seeds, idx); P = orc.patches_from_seeds(seeds, idx)
    x0 = np.array([orc.initial_parameters(osc, P[k]) for k in range(len(idx))])
    rng = np.random.default_rng(5)
    for trial in range(2):
        x = x0 + (rng.uniform(-1, 1, size=x0.shape) * np.array([0.3, 2.0, 2.0]) if trial else 0.0)
        fg, gg = api.objective_batch(g, batch, x)
        fc = np.array([orc.objective_at(osc, P[k], x[k]) for k in range(len(idx))])
        print('  objective exact-equal %.4f' % np.mean(fg == fc), 'max |d| %.3g' % np.abs(fg - fc).max(), 'mean grabs', gg.mean())
    for ref, rob in [(0, 0), (2, 1)]:
        inc = api.inccs_batch(g, batch, ref, rob)
        eq = [np.array_equal(inc[k, :P[k].n_images], orc.inccs(osc, P[k], ref, rob)) for k in range(len(idx))]
        print('  inccs(ref=%d,robust=%d) exact-equal %.4f' % (ref, rob, np.mean(eq)))
    b = api.Batch.from_seeds(seeds); api.optimize_batch(g, b)
    Q = orc.patches_from_seeds(seeds); orc.optimize_batch(osc, Q, n_threads=16)
    okc = np.array([q.stage == 0 for q in Q]); okg = b.ok.astype(bool)
    both = np.nonzero(okc & okg)[0]
    same = [np.array_equal(np.array(Q[k].center[:], dtype=np.float32), b.center[k]) and np.array_equal(np.array(Q[k].normal[:], dtype=np.float32), b.normal[k]) and list(Q[k].images[:Q[k].n_images]) == list(b.images[k, :b.n_images[k]]) and np.array_equal(np.array(Q[k].color[:], dtype=np.float32), b.color[k]) and Q[k].nevals == b.nevals[k] for k in both]
    print('  optimize: stages equal %.4f' % np.mean(np.array([q.stage for q in Q]) == b.stage), 'both ok', len(both), 'bit-identical %.4f' % np.mean(same), 'final n_images max', b.n_images.max())
    for k in [k for k, s in zip(both, same) if not s][:4]:
        print('   mismatch', k, 'nevals', Q[k].nevals, b.nevals[k], 'x', list(Q[k].x), b.x[k].tolist(), 'nimg', Q[k].n_images, b.n_images[k])
