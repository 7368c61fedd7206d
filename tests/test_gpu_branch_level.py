"""SURVEY 8f-3, the BRANCH half as a composed level (round 6): CellProcessor::branch (reference
src/hpmvs/CellProcessor.cpp:210-307) splits a leaf -- level-support gate (:221-224, Scene::getLevelSupport, Scene.cpp:334-343),
four diagonal children with Cell::contains before and after optimize (:233-258), and then the depth maps: the leaf's patch is
taken back with Scene::setDepths(patch, subtract = true) (:276-279, Scene.cpp:373-374) before the children are entered (:296).
Subtraction does not commute with the minimum of an ordinary setDepths, so a batch of such calls has to end with the maps of the
reference's SEQUENTIAL loop: hpmvs_depth_ops_batch (keys, sort, per-cell replay in call order), hpmvs_level_support_batch and
hpmvs_amd.frontier.branch_level against the oracle's leaf-by-leaf restatement (orc_set_depths_ex, orc_level_support,
orc_branch_round) -- every cell of every map, every child, every split decision."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_patches(batch, idx=None):
    from oracle import oracle as orc
    idx = range(batch.n) if idx is None else idx
    arr = (orc.Patch * len(idx))()
    for j, k in enumerate(idx):
        p = arr[j]
        p.center[:] = batch.center[k].tolist(); p.normal[:] = batch.normal[k].tolist()
        p.scale = float(batch.scale[k])
        p.n_images = int(batch.n_images[k])
        for i in range(p.n_images):
            p.images[i] = int(batch.images[k, i])
    return arr


def _maps_equal(gpu_scene, OD, n_views, n_levels):
    from hpmvs_amd import api
    for v in range(n_views):
        for l in range(n_levels):
            a, b = api.depth_level(gpu_scene, v, l), OD.level(v, l)
            if not np.array_equal(a, b):
                return False, (v, l, int((a != b).sum()))
    return True, None


def _refined(scene, gscene, n, seed_off=21):
    from hpmvs_amd import api, synth
    seeds = synth.make_seeds(scene, n, start_level=2, seed=synth.SEED + seed_off)
    b = api.Batch.from_seeds(seeds)
    api.optimize_batch(gscene, b)
    keep = np.nonzero(b.ok)[0]
    assert len(keep) > n // 2
    return api.Batch(b.center[keep], b.normal[keep], b.scale[keep], b.n_images[keep], b.images[keep])


def test_ordered_depth_ops_equal_the_sequential_loop(tiny_scene, oracle_scene, gpu_scene):
    """A long random sequence of setDepths(patch, subtract) calls -- the same patches entered, taken back and entered again,
    and copies pushed along a viewing ray so that DIFFERENT depths meet in ONE cell (where the order decides the result) --
    as one hpmvs_depth_ops_batch: every cell of every map equals the loop that makes the calls one after the other."""
    from hpmvs_amd import api
    from oracle import oracle as orc
    R = _refined(tiny_scene, gpu_scene, 1200)
    rng = np.random.default_rng(5)
    n0 = R.n
    # copies of every patch a little nearer to / farther from the camera of its first image: same pixel there, another depth
    cams = np.array([tiny_scene.views[int(R.images[k, 0])].c for k in range(n0)], dtype=np.float64)
    fac = rng.choice([0.999, 0.9995, 1.0005, 1.001], size=n0)
    moved = (cams + (R.center[:, :3].astype(np.float64) - cams) * fac[:, None]).astype(np.float32)
    center2 = R.center.copy(); center2[:, :3] = moved
    pool_c = np.concatenate([R.center, center2]); pool_n = np.concatenate([R.normal, R.normal])
    pool_s = np.concatenate([R.scale, R.scale]); pool_m = np.concatenate([R.n_images, R.n_images]); pool_i = np.concatenate([R.images, R.images])
    # the sequence: every original entered; then 6000 random calls (40 % subtractions) over originals and copies
    seq = list(range(n0)) + rng.integers(0, 2 * n0, size=6000).tolist()
    sub = np.zeros(len(seq), np.uint8)
    sub[n0:] = (rng.random(len(seq) - n0) < 0.4).astype(np.uint8)
    idx = np.array(seq)
    ops = api.Batch(pool_c[idx], pool_n[idx], pool_s[idx], pool_m[idx], pool_i[idx])
    ops.ok[:] = 1
    ops.ok[rng.integers(n0, len(seq), size=200)] = 0   # (calls that are not made: ok == 0)
    api.depth_reset(gpu_scene)
    api.depth_ops_batch(gpu_scene, ops, sub)
    OD = orc.OracleDepths(oracle_scene)
    P = _oracle_patches(ops)
    for t in range(ops.n):
        if ops.ok[t]:
            OD.set_depths(P[t], subtract=bool(sub[t]))
    ok, where = _maps_equal(gpu_scene, OD, tiny_scene.n_views, oracle_scene.camera(0).n_levels)
    assert ok, where
    # the sequence did meet cells with several different depths, and subtractions did clear cells
    n_set = sum(int((OD.level(v, l) < 1000.0).sum()) for v in range(tiny_scene.n_views) for l in range(oracle_scene.camera(0).n_levels))
    OD2 = orc.OracleDepths(oracle_scene)
    for t in range(ops.n):
        if ops.ok[t] and not sub[t]:
            OD2.set_depths(P[t])
    n_set_without = sum(int((OD2.level(v, l) < 1000.0).sum()) for v in range(tiny_scene.n_views) for l in range(oracle_scene.camera(0).n_levels))
    assert 0 < n_set < n_set_without
    same_as_unordered = all(np.array_equal(OD.level(v, l), OD2.level(v, l)) for v in range(tiny_scene.n_views) for l in range(oracle_scene.camera(0).n_levels))
    assert not same_as_unordered
    # subtract == all zeros is the plain minimum (hpmvs_set_depths_batch), in any order
    api.depth_reset(gpu_scene)
    api.depth_ops_batch(gpu_scene, ops, np.zeros(ops.n, np.uint8))
    OD3 = orc.OracleDepths(oracle_scene)
    for t in range(ops.n):
        if ops.ok[t]:
            OD3.set_depths(P[t])
    ok3, where3 = _maps_equal(gpu_scene, OD3, tiny_scene.n_views, oracle_scene.camera(0).n_levels)
    assert ok3, where3


def test_level_support_matches_the_oracle(tiny_scene, oracle_scene, gpu_scene):
    """Scene::getLevelSupport on patches whose scale runs over two octaves around the level boundaries, MINLEVEL 0..3."""
    from hpmvs_amd import api
    from oracle import oracle as orc
    R = _refined(tiny_scene, gpu_scene, 600)
    rng = np.random.default_rng(9)
    R.scale[:] = (R.scale * np.exp2(rng.uniform(-2.5, 2.5, size=R.n))).astype(np.float32)
    P = _oracle_patches(R)
    seen = set()
    for m in range(4):
        got = api.level_support_batch(gpu_scene, R, m)
        want = np.array([orc.level_support(oracle_scene, P[k], m) for k in range(R.n)])
        assert np.array_equal(got, want), (m, np.nonzero(got != want)[0][:10])
        seen.update(np.unique(got).tolist())
    assert 0 in seen and max(seen) >= 2


def _branch_against_the_loop(tag, scene, gscene, oscene, n_seeds):
    from hpmvs_amd import api, frontier
    from oracle import oracle as orc
    R = _refined(scene, gscene, n_seeds, seed_off=33)
    n = R.n
    rng = np.random.default_rng(13)
    # a quarter of the leaves on a coarse scale (no level support: exhausted), the leaf centred on its patch as extend leaves it
    R.scale[rng.random(n) < 0.25] *= np.float32(0.2)
    width = (R.scale * np.float32(2.0 / 0.9)).astype(np.float32)
    cc = R.center[:, :3].copy()
    off = rng.random(n) < 0.1   # leaves whose patch lies outside them: no child passes Cell::contains (:247) -- nothing is refined
    cc[off] += 3.0 * width[off, None]
    final = (rng.random(n) < 0.5).astype(np.uint8)
    n_levels = oscene.camera(0).n_levels
    # the state before the level: every leaf's patch is in the maps
    api.depth_reset(gscene)
    R.ok[:] = 1
    api.set_depths_batch(gscene, R)
    OD = orc.OracleDepths(oscene)
    P = _oracle_patches(R)
    for k in range(n):
        OD.set_depths(P[k])
    ok, where = _maps_equal(gscene, OD, scene.n_views, n_levels)
    assert ok, where
    res = frontier.branch_level(gscene, R, cc, width, final)
    cand, split = orc.branch_round(oscene, OD, P, cc, width, final, which=orc.OPT_REF)
    st = np.array([cand[t].stage for t in range(4 * n)])
    sup = np.array([orc.level_support(oscene, P[k], 0) for k in range(n)])
    assert np.array_equal(res.support, sup)
    # stages: 27 (the oracle's "no level support: nothing built") is the batch's 20 (skipped)
    assert np.array_equal(np.where(st == 27, 20, st), res.candidates.stage), (tag, np.nonzero(np.where(st == 27, 20, st) != res.candidates.stage)[0][:10])
    assert np.array_equal(split, res.split), (tag, np.nonzero(split != res.split)[0][:10])
    for t in np.nonzero(st == 0)[0]:
        assert np.array_equal(np.array(cand[t].center[:], dtype=np.float32), res.candidates.center[t]), (tag, t)
        assert np.array_equal(np.array(cand[t].normal[:], dtype=np.float32), res.candidates.normal[t]), (tag, t)
        m = cand[t].n_images
        assert m == res.candidates.n_images[t] and list(cand[t].images[:m]) == list(res.candidates.images[t, :m]), (tag, t)
    ok, where = _maps_equal(gscene, OD, scene.n_views, n_levels)
    assert ok, (tag, where)
    rep = {"scene": tag, "leaves": n, "without_level_support": int((sup < 1).sum()), "split": int(split.sum()),
           "kept_on_the_final_level": int(((sup >= 1) & (split == 0)).sum()), "children": int((st == 0).sum())}
    print("branch level:", rep)
    assert rep["without_level_support"] > 0 and rep["split"] > 10 and rep["children"] > 10 and rep["kept_on_the_final_level"] > 0
    return rep


def test_branch_level_equals_the_sequential_loop_on_configs0(tiny_scene, oracle_scene, gpu_scene):
    _branch_against_the_loop("configs0_3v_640x480", tiny_scene, gpu_scene, oracle_scene, 400)


def test_branch_level_equals_the_sequential_loop_on_a_12_view_scene():
    from hpmvs_amd import api, synth
    from oracle import oracle as orc
    scene = synth.make_scene(12, 640, 480, n_waves=24)
    g = api.Scene(scene, device=0)
    try:
        _branch_against_the_loop("12v_640x480", scene, g, orc.OracleScene(scene), 700)
    finally:
        g.close()
