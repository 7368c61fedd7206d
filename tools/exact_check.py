"""How exact is the GPU path vs the oracle?  (objective values, INCCs, full optimize)"""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from hpmvs_amd import api, synth
from oracle import oracle as orc
scene = synth.make_scene(3, 640, 480, n_waves=24)
g = api.Scene(scene); osc = orc.OracleScene(scene)
for sl in (2, 3):
    seeds = synth.make_seeds(scene, 300, start_level=sl, seed=synth.SEED + sl)
    idx = np.nonzero(seeds.n_images >= 3)[0]
    batch = api.Batch.from_seeds(seeds, idx); P = orc.patches_from_seeds(seeds, idx)
    x0 = np.array([orc.initial_parameters(osc, P[k]) for k in range(len(idx))])
    rng = np.random.default_rng(5)
    for trial in range(3):
        x = x0 + (rng.uniform(-1, 1, size=x0.shape) * np.array([0.3, 2.0, 2.0]) if trial else 0.0)
        fg, _ = api.objective_batch(g, batch, x)
        fc = np.array([orc.objective_at(osc, P[k], x[k]) for k in range(len(idx))])
        print('level', sl, 'trial', trial, 'objective: exact-equal fraction %.4f' % np.mean(fg == fc), 'max |d| %.3g' % np.abs(fg - fc).max())
    inc = api.inccs_batch(g, batch, 0, 1)
    eq = [np.array_equal(inc[k, :P[k].n_images], orc.inccs(osc, P[k], 0, 1)) for k in range(len(idx))]
    print('level', sl, 'inccs exact-equal fraction %.4f' % np.mean(eq))
    b = api.Batch.from_seeds(seeds); api.optimize_batch(g, b)
    Q = orc.patches_from_seeds(seeds); orc.optimize_batch(osc, Q, n_threads=8)
    okc = np.array([q.stage == 0 for q in Q]); okg = b.ok.astype(bool)
    both = np.nonzero(okc & okg)[0]
    same_x = [np.array_equal(np.array(Q[k].x[:]), b.x[k]) for k in both]
    same_all = [np.array_equal(np.array(Q[k].center[:]), b.center[k]) and np.array_equal(np.array(Q[k].normal[:]), b.normal[k]) and list(Q[k].images[:Q[k].n_images]) == list(b.images[k, :b.n_images[k]]) and np.array_equal(np.array(Q[k].color[:]), b.color[k]) for k in both]
    print('level', sl, 'optimize: decisions equal %.4f' % np.mean(okc == okg), 'both ok', len(both), 'bit-identical x %.4f' % np.mean(same_x), 'bit-identical patch (center,normal,images,color) %.4f' % np.mean(same_all), 'nevals equal %.4f' % np.mean([Q[k].nevals == b.nevals[k] for k in both]))
    bad = [k for k, s in zip(both, same_x) if not s][:5]
    for k in bad:
        print('   mismatch patch', k, 'x cpu', list(Q[k].x), 'x gpu', b.x[k].tolist(), 'nevals', Q[k].nevals, b.nevals[k])
