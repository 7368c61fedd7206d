"""GPU parity of the photometric kernels against the CPU oracle (through the C ABI).

Tolerance: NONE.  The kernels execute the reference's float32 arithmetic operation for operation
(sequential summation order of Patch2d::normalize/dot, incremental sample stepping, no FMA
contraction), so pyramid bytes, setINCCs entries and objective_fn values must be BIT-IDENTICAL to
the CPU oracle (SURVEY 8d had proposed |d| <= 1e-4).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pyramid_bit_exact(tiny_scene, oracle_scene, gpu_scene):
    for view in range(tiny_scene.n_views):
        for level in range(6):
            a = oracle_scene.level(view, level)
            b = gpu_scene.level(view, level)
            assert a.shape == b.shape
            assert np.array_equal(a, b), f"view {view} level {level}: {(a != b).sum()} bytes differ"


def test_pyramid_odd_sizes():
    from hpmvs_amd import api
    from oracle import oracle as orc
    rng = np.random.default_rng(7)
    for (h, w) in [(2, 2), (3, 5), (7, 9), (33, 17), (101, 257), (480, 641)]:
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        assert np.array_equal(orc.half_resize(img), api.build_pyramid(img)), (h, w)


def test_inccs_parity(tiny_seeds, oracle_scene, gpu_scene):
    from hpmvs_amd import api
    from oracle import oracle as orc
    idx = np.nonzero(tiny_seeds.n_images >= 2)[0]
    batch = api.Batch.from_seeds(tiny_seeds, idx)
    P = orc.patches_from_seeds(tiny_seeds, idx)
    for ref_idx, robust in [(0, 0), (0, 1), (1, 1)]:
        got = api.inccs_batch(gpu_scene, batch, ref_idx, robust)
        worst = 0.0
        for k in range(len(idx)):
            want = orc.inccs(oracle_scene, P[k], ref_idx, robust)
            n = P[k].n_images
            worst = max(worst, float(np.abs(got[k, :n] - want).max()))
        assert worst == 0.0, (ref_idx, robust, worst)


def test_objective_parity(tiny_seeds, oracle_scene, gpu_scene):
    from hpmvs_amd import api
    from oracle import oracle as orc
    idx = np.nonzero(tiny_seeds.n_images >= 3)[0]
    batch = api.Batch.from_seeds(tiny_seeds, idx)
    P = orc.patches_from_seeds(tiny_seeds, idx)
    rng = np.random.default_rng(11)
    x0 = np.array([orc.initial_parameters(oracle_scene, P[k]) for k in range(len(idx))])
    for trial in range(4):
        x = x0 + (rng.uniform(-1, 1, size=x0.shape) * np.array([0.3, 2.0, 2.0]) if trial else 0.0)
        f_gpu, g_gpu = api.objective_batch(gpu_scene, batch, x)
        f_cpu = np.array([orc.objective_at(oracle_scene, P[k], x[k]) for k in range(len(idx))])
        assert np.array_equal(f_gpu, f_cpu), np.abs(f_gpu - f_cpu).max()
        assert (f_cpu < 2.0).sum() > 10


@pytest.mark.parametrize("kind", ["noise", "flat_blocks", "saturated"])
def test_photometric_parity_on_hostile_images(kind, tiny_scene, tiny_seeds):
    """Images the smooth synthetic texture never produces: uint8 noise (large variances), 16x16 constant blocks
    (windows with zero variance -> the `ave2 == 0 -> 1` branch of Patch2d::normalize, Patch2d.hpp:71-72) and
    0/255 saturation.  setINCCs, objective_fn and the full refinement stay bit-identical to the oracle; the
    packed evaluator (optimize) and the one-request evaluator (objective/inccs kernels) both see them."""
    import copy
    from hpmvs_amd import api
    from oracle import oracle as orc
    rng = np.random.default_rng({"noise": 1, "flat_blocks": 2, "saturated": 3}[kind])
    scene = copy.copy(tiny_scene)
    scene.views = []
    for v in tiny_scene.views:
        h, w = v.height, v.width
        if kind == "noise":
            img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        elif kind == "flat_blocks":
            small = rng.integers(0, 256, size=(h // 16 + 1, w // 16 + 1, 3), dtype=np.uint8)
            img = np.repeat(np.repeat(small, 16, axis=0), 16, axis=1)[:h, :w]
        else:
            img = (rng.integers(0, 2, size=(h // 4 + 1, w // 4 + 1, 3), dtype=np.uint8) * 255)
            img = np.repeat(np.repeat(img, 4, axis=0), 4, axis=1)[:h, :w]
        nv = copy.copy(v)
        nv.rgb = np.ascontiguousarray(img)
        scene.views.append(nv)
    g = api.Scene(scene)
    osc = orc.OracleScene(scene)
    idx = np.nonzero(tiny_seeds.n_images >= 3)[0][:120]
    batch = api.Batch.from_seeds(tiny_seeds, idx)
    P = orc.patches_from_seeds(tiny_seeds, idx)
    for ref_idx, robust in [(0, 0), (1, 1)]:
        got = api.inccs_batch(g, batch, ref_idx, robust)
        for k in range(len(idx)):
            assert np.array_equal(got[k, :P[k].n_images], orc.inccs(osc, P[k], ref_idx, robust)), (kind, k)
    x0 = np.array([orc.initial_parameters(osc, P[k]) for k in range(len(idx))])
    f_gpu, _ = api.objective_batch(g, batch, x0)
    f_cpu = np.array([orc.objective_at(osc, P[k], x0[k]) for k in range(len(idx))])
    assert np.array_equal(f_gpu, f_cpu)
    b = api.Batch.from_seeds(tiny_seeds, idx)
    api.optimize_batch(g, b)
    Q = orc.patches_from_seeds(tiny_seeds, idx)
    orc.optimize_batch(osc, Q, n_threads=8)
    assert np.array_equal(np.array([q.stage for q in Q]), b.stage)
    for k in np.nonzero(b.ok)[0]:
        assert np.array_equal(np.array(Q[k].center[:], dtype=np.float32), b.center[k])
        assert Q[k].nevals == b.nevals[k]
    g.close()


def test_pyramid_matches_the_genuine_cimg():
    """half_resize_kernel against the reference's vendored CImg: the committed goldens
    (tests/golden/g4_cimg_pyramid.npz, produced by CImg itself) and, when oracle/_ref/libref_cimg.so
    travelled, the library on fresh random images -- whole pyramids, byte for byte."""
    import os
    from hpmvs_amd import api
    from oracle import oracle as orc
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g4_cimg_pyramid.npz"))
    for name in sorted({k.split("_")[0] for k in z.files}):
        cur, l = z[f"{name}_in"], 1
        while f"{name}_l{l}" in z.files:
            cur = api.build_pyramid(cur)
            assert np.array_equal(cur, z[f"{name}_l{l}"]), (name, l)
            l += 1
    assert orc.cimg_available(), "oracle/_ref/libref_cimg.so did not travel"
    rng = np.random.default_rng(5)
    for (h, w) in [(9, 7), (64, 33), (480, 640), (1080, 1920), (2160, 3840)]:
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        ref = orc.cimg_pyramid(img, 6 if min(h, w) >= 64 else 3)
        cur = img
        for l in range(1, len(ref)):
            cur = api.build_pyramid(cur)
            assert np.array_equal(cur, ref[l]), (h, w, l)


def test_level_choice_in_the_float_rounding_band_of_log2(tiny_scene, tiny_seeds, oracle_scene, gpu_scene):
    """Camera::getLeveli (Camera.cpp:92-99) rounds log2(r) to float BEFORE std::round: an r a hair below 2^(k+1/2) whose
    logarithm rounds up to exactly k + 0.5f gets level k + 1, not k.  The device counts thresholds instead of taking a
    logarithm; its thresholds are the smallest doubles for which the host's (float)log2 reaches k + 0.5 (DevScene::lvl_thr),
    so it must agree inside that band too (VERDICT r3 asked for the edge to be argued or tested).  For a few patches and
    each of their images the patch scale walks, float by float, across 2^(1.5) and 2^(2.5): setINCCs on the GPU equals the
    oracle (whose cam_get_level is the reference's expression) at every one of them, and the walk really contains scales
    inside the band (where comparing r with 2^(k+1/2) itself would have chosen the other level)."""
    from hpmvs_amd import api
    from oracle import oracle as orc
    f32 = np.float32
    idx = np.nonzero(tiny_seeds.n_images >= 3)[0][:3]
    in_band = 0
    for p in idx:
        c = tiny_seeds.center[p].astype(f32)
        scales = []
        for j in range(int(tiny_seeds.n_images[p])):
            cam = oracle_scene.camera(int(tiny_seeds.images[p, j]))
            d = c - np.array(cam.center[:], dtype=f32)
            fz = np.sqrt(f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2])) + f32(d[3] * d[3]))
            fsum = f32(f32(cam.K0[0]) + f32(cam.K0[4]))
            for k in (1, 2):
                s0 = f32(2.0 ** (k + 0.5) * 2.0 * float(fz) / float(fsum))
                s = np.nextafter(s0, f32(0), dtype=f32)
                for _ in range(60):
                    s = np.nextafter(s, f32(0), dtype=f32)
                for _ in range(120):
                    s = np.nextafter(s, f32(np.inf), dtype=f32)
                    r = float(f32(s * fsum)) / (2.0 * float(fz))
                    lvl_ref = int(np.round(f32(np.log2(r))))
                    lvl_naive = sum(1 for kk in range(5) if r >= 2.0 ** (kk + 0.5))
                    in_band += int(lvl_ref != lvl_naive)
                    scales.append(s)
        scales = np.array(scales, dtype=f32)
        n = len(scales)
        rep = np.full(n, p)
        batch = api.Batch(tiny_seeds.center[rep], tiny_seeds.normal[rep], scales, tiny_seeds.n_images[rep], tiny_seeds.images[rep])
        got = api.inccs_batch(gpu_scene, batch, 0, 0)
        P = orc.patches_from_seeds(tiny_seeds, rep)
        m = int(tiny_seeds.n_images[p])
        for t in range(n):
            P[t].scale = float(scales[t])
            want = orc.inccs(oracle_scene, P[t], 0, 0)
            assert np.array_equal(got[t, :m], want), (p, t, float(scales[t]), got[t, :m], want)
    assert in_band >= 1, "the walk never entered the band: the test would not see the edge"
