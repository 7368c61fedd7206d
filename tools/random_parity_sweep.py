"""Confidence sweep: random synthetic scenes (view count, resolution, texture frequency, start level, seed
displacement) and -- with a second argument `opts` -- random HpmvsOptions (levels, angles, image counts, NCC thresholds)
-> GPU refinement vs the CPU oracle, patch by patch.   random_parity_sweep.py <seed> [opts] [scenes]
(tools/mismatch_diag.py replays one scene of a sweep and says what a mismatching patch's difference starts with.)"""
import sys, numpy as np
sys.path.insert(0, '.')


def draw(rng, random_opts):
    """the random choices of one scene, in the order the sweep has always drawn them"""
    c = dict(nv=int(rng.integers(3, 17)), W=int(rng.choice([320, 512, 640, 800, 1024])))
    c['H'] = int(c['W'] * rng.choice([0.5625, 0.75]))
    c['waves'] = int(rng.integers(6, 28)); c['sl'] = int(rng.integers(1, 4)); c['n'] = 400
    c['max_seed_images'] = int(rng.integers(3, 12)); c['seed'] = int(rng.integers(1, 1 << 30)); c['displace'] = float(rng.uniform(0.1, 1.5))
    c['opts'] = None
    if random_opts:
        c['opts'] = dict(MAXLEVEL=int(rng.integers(2, 6)), MINLEVEL=int(rng.integers(0, 2)),
                         MAX_ANGLE=float(np.float32(np.deg2rad(rng.choice([45.0, 60.0, 75.0])))),
                         MIN_ANGLE=float(np.float32(np.deg2rad(rng.choice([0.0, 3.0, 10.0])))),
                         MAX_IMAGES_PER_PATCH=int(rng.choice([4, 8, 12, 32])), MIN_IMAGES_PER_PATCH=int(rng.integers(2, 5)),
                         NCC_ALPHA_1=float(rng.choice([0.2, 0.3, 0.4, 0.6])), NCC_ALPHA_2=float(rng.choice([0.3, 0.5, 0.7, 0.9])))
    return c


def run_scene(c, dev):
    """-> (gpu batch, oracle patches, seeds, scene, indices of the patches that differ)"""
    from hpmvs_amd import api, synth
    from oracle import oracle as orc
    scene = synth.make_scene(c['nv'], c['W'], c['H'], n_waves=c['waves'], device=dev)
    seeds = synth.make_seeds(scene, c['n'], start_level=c['sl'], max_images=32, max_seed_images=c['max_seed_images'],
                             seed=c['seed'], displace=c['displace'])
    g = api.Scene(scene); osc = orc.OracleScene(scene)
    og, oc = api.default_options(), orc.default_options()
    for k, v in (c['opts'] or {}).items(): setattr(og, k, v); setattr(oc, k, v)
    b = api.Batch.from_seeds(seeds); api.optimize_batch(g, b, og)
    P = orc.patches_from_seeds(seeds); orc.optimize_batch(osc, P, options=oc, n_threads=16)
    st = np.array([p.stage for p in P])
    bad = set(np.nonzero(st != b.stage)[0].tolist())
    for k in np.nonzero(b.ok)[0]:
        same = (np.array_equal(np.array(P[k].center[:], dtype=np.float32), b.center[k]) and
                np.array_equal(np.array(P[k].normal[:], dtype=np.float32), b.normal[k]) and
                list(P[k].images[:P[k].n_images]) == list(b.images[k, :b.n_images[k]]) and P[k].nevals == b.nevals[k])
        if not same: bad.add(int(k))
    g.close()
    return b, P, seeds, scene, sorted(bad)


if __name__ == '__main__':
    import torch
    dev = torch.device('cuda', 0)
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    RANDOM_OPTS = len(sys.argv) > 2 and sys.argv[2] == 'opts'
    N_SCENES = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    tot = bad = okc = 0
    for it in range(N_SCENES):
        c = draw(rng, RANDOM_OPTS)
        b, P, seeds, scene, mism = run_scene(c, dev)
        desc = (' opts ' + ' '.join('%s=%g' % (k, v) for k, v in c['opts'].items())) if c['opts'] else ''
        tot += c['n']; bad += len(mism); okc += int(b.ok.sum())
        print('scene', it, 'views', c['nv'], '%dx%d' % (c['W'], c['H']), 'start_level', c['sl'], 'refined', int(b.ok.sum()), 'mismatches', len(mism), mism[:4], desc)
    print('TOTAL patches', tot, 'refined', okc, 'mismatching', bad)
