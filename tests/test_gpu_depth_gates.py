"""SURVEY 8f-3, second half: the depth maps and the three acceptance counts of the expansion
(Scene::setDepths, depthTests, viewBlockTest, pixelFreeTests -- reference src/hpmvs/Scene.cpp:351-381, 518-644;
callers src/hpmvs/CellProcessor.cpp:134-142, 198-200) through the C ABI against the oracle's restatement.
Integer counts and float minima: identical, for both readings of the unqualified abs() at Scene.cpp:571."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _refined(tiny_scene, gpu_scene, n=1500):
    from hpmvs_amd import api, synth
    seeds = synth.make_seeds(tiny_scene, n, start_level=2, seed=synth.SEED + 11)
    batch = api.Batch.from_seeds(seeds)
    api.optimize_batch(gpu_scene, batch)
    assert batch.ok.sum() > n // 2
    return seeds, batch


def _oracle_patches(batch):
    from oracle import oracle as orc
    arr = (orc.Patch * batch.n)()
    for k in range(batch.n):
        p = arr[k]
        p.center[:] = batch.center[k].tolist(); p.normal[:] = batch.normal[k].tolist()
        p.scale = float(batch.scale[k])
        p.n_images = int(batch.n_images[k])
        for j in range(p.n_images):
            p.images[j] = int(batch.images[k, j])
    return arr


def test_set_depths_and_gates_match_the_oracle(tiny_scene, oracle_scene, gpu_scene):
    from hpmvs_amd import api
    from oracle import oracle as orc
    seeds, batch = _refined(tiny_scene, gpu_scene)
    P = _oracle_patches(batch)
    ok = batch.ok.astype(bool)
    D = orc.OracleDepths(oracle_scene)
    api.depth_reset(gpu_scene)
    n_levels = oracle_scene.camera(0).n_levels
    # empty maps first: every in-bounds pixel is free, nothing blocks, similar-depth passes wherever the 3x3 block is inside
    v0, b0, f0 = api.depth_gates_batch(gpu_scene, batch, 1.0, 0)
    for k in range(0, batch.n, 7):
        assert (v0[k], b0[k], f0[k]) == D.gates(P[k], 1.0, 0), k
    assert b0.max() == 0 and v0[ok].min() >= 0 and f0[ok].max() <= batch.n_images[ok].max()
    # record the first half of the refined patches (the rest stays unrecorded: candidates to be judged)
    first = ok & (np.arange(batch.n) < batch.n // 2)
    for k in np.nonzero(first)[0]:
        D.set_depths(P[k])
    keep = batch.ok.copy()
    batch.ok[:] = first.astype(np.uint8)
    api.set_depths_batch(gpu_scene, batch)
    batch.ok[:] = keep
    written = 0
    for view in range(tiny_scene.n_views):
        for level in range(n_levels):
            g, o = api.depth_level(gpu_scene, view, level), D.level(view, level)
            assert g.shape == o.shape, (view, level)
            assert np.array_equal(g, o), (view, level, int((g != o).sum()))
            written += int((g < 1000.0).sum())
    assert written > 100
    # the gates over the written maps, both readings of abs()
    for abs_int in (0, 1):
        v, b, f = api.depth_gates_batch(gpu_scene, batch, 1.0, abs_int)
        want = np.array([D.gates(P[k], 1.0, abs_int) for k in range(batch.n)])
        assert np.array_equal(v, want[:, 0]) and np.array_equal(b, want[:, 1]) and np.array_equal(f, want[:, 2]), abs_int
        # a recorded patch agrees with its own depth (diff = 0 at the centre pixel's cell or a nearer surface)
        assert (f[first] < batch.n_images[first]).any()
    # a tighter margin can only lower the similar-depth count
    v_tight, _, _ = api.depth_gates_batch(gpu_scene, batch, 0.05, 0)
    assert np.all(v_tight <= v0) and (v_tight < v0).any()
    # the maps survive a host round trip (the scheduler's own matrices copied in as they are)
    m = api.depth_level(gpu_scene, 0, 1)
    m2 = m.copy(); m2[3, 2] = 7.5
    api._chk(api.lib().hpmvs_scene_depth_set_level(gpu_scene.h, 0, 1, m2.ctypes.data, m2.shape[1], m2.shape[0]))
    assert np.array_equal(api.depth_level(gpu_scene, 0, 1), m2)


def test_depth_entry_points_need_reset_first(tiny_scene):
    from hpmvs_amd import api, synth
    g = api.Scene(tiny_scene, device=0)
    seeds = synth.make_seeds(tiny_scene, 8, start_level=2)
    b = api.Batch.from_seeds(seeds)
    with pytest.raises(api.HpmvsError):
        api.depth_gates_batch(g, b)
    with pytest.raises(api.HpmvsError):
        api.depth_level(g, 0, 0)
    api.depth_reset(g)
    assert np.all(api.depth_level(g, 0, 0) == np.float32(1000.0))
    g.close()
