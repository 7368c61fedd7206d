"""Random synthetic scenes -- view count, resolution, aspect, texture bandwidth, START_LEVEL, seed displacement, number of
attached images per seed -- refined on the GPU and by the oracle, patch by patch (tools/random_parity_sweep.py as a test, with a
fixed seed per scene so that a failure names a reproducible case).  Bit for bit: failing stage of every patch; centre, normal,
image list and evaluation count of every refined one.  The scenes are small (400 seeds) so that the oracle needs seconds."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scene_seed", [101, 202, 303, 404, 505, 606])
def test_random_scene_refines_bit_identically(scene_seed):
    import os
    import torch
    from hpmvs_amd import api, synth
    from oracle import oracle as orc
    if api.device_count() < 1:
        pytest.fail("no HIP device: -m gpu tests need the MI355X box (no CPU fallback exists)")
    rng = np.random.default_rng(scene_seed)
    nv = int(rng.integers(3, 17))
    W = int(rng.choice([320, 512, 640, 800, 1024]))
    H = int(W * rng.choice([0.5625, 0.75]))
    waves = int(rng.integers(6, 28))
    sl = int(rng.integers(1, 4))
    n = 400
    scene = synth.make_scene(nv, W, H, n_waves=waves, device=torch.device("cuda", 0))
    seeds = synth.make_seeds(scene, n, start_level=sl, max_images=32, max_seed_images=int(rng.integers(3, 12)),
                             seed=int(rng.integers(1, 1 << 30)), displace=float(rng.uniform(0.1, 1.5)))
    g = api.Scene(scene)
    b = api.Batch.from_seeds(seeds)
    api.optimize_batch(g, b)
    P = orc.patches_from_seeds(seeds)
    orc.optimize_batch(orc.OracleScene(scene), P, which=orc.best_optimizer(), n_threads=min(16, os.cpu_count() or 1))
    tag = dict(scene_seed=scene_seed, views=nv, size=f"{W}x{H}", start_level=sl, refined=int(b.ok.sum()))
    assert np.array_equal(np.array([p.stage for p in P]), b.stage), tag
    for k in np.nonzero(b.ok)[0]:
        assert np.array_equal(np.array(P[k].center[:], dtype=np.float32), b.center[k]), (tag, k)
        assert np.array_equal(np.array(P[k].normal[:], dtype=np.float32), b.normal[k]), (tag, k)
        assert list(P[k].images[:P[k].n_images]) == list(b.images[k, :b.n_images[k]]) and P[k].nevals == b.nevals[k], (tag, k)
    print(tag)
    g.close()
