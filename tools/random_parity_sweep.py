"""One-off confidence sweep: random synthetic scenes (view count, resolution, texture frequency, start level,
seed displacement) -> GPU refinement vs the CPU oracle, patch by patch."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from hpmvs_amd import api, synth
from oracle import oracle as orc
dev = torch.device('cuda', 0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
tot = bad = okc = 0
for it in range(12):
    nv = int(rng.integers(3, 17)); W = int(rng.choice([320, 512, 640, 800, 1024])); H = int(W * rng.choice([0.5625, 0.75]))
    waves = int(rng.integers(6, 28)); sl = int(rng.integers(1, 4)); n = 400
    scene = synth.make_scene(nv, W, H, n_waves=waves, device=dev)
    seeds = synth.make_seeds(scene, n, start_level=sl, max_images=32, max_seed_images=int(rng.integers(3, 12)),
                             seed=int(rng.integers(1, 1 << 30)), displace=float(rng.uniform(0.1, 1.5)))
    g = api.Scene(scene); osc = orc.OracleScene(scene)
    b = api.Batch.from_seeds(seeds); api.optimize_batch(g, b)
    P = orc.patches_from_seeds(seeds); orc.optimize_batch(osc, P, n_threads=16)
    st = np.array([p.stage for p in P])
    mism = int((st != b.stage).sum())
    for k in np.nonzero(b.ok)[0]:
        same = (np.array_equal(np.array(P[k].center[:], dtype=np.float32), b.center[k]) and
                np.array_equal(np.array(P[k].normal[:], dtype=np.float32), b.normal[k]) and
                list(P[k].images[:P[k].n_images]) == list(b.images[k, :b.n_images[k]]) and P[k].nevals == b.nevals[k])
        mism += (not same)
    tot += n; bad += mism; okc += int(b.ok.sum())
    print('scene', it, 'views', nv, '%dx%d' % (W, H), 'start_level', sl, 'refined', int(b.ok.sum()), 'mismatches', mism)
    g.close()
print('TOTAL patches', tot, 'refined', okc, 'mismatching', bad)
