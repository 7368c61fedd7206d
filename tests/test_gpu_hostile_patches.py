"""Non-finite and degenerate PATCHES inside an ordinary batch (the reference's callers cannot produce them: an NVM point is
finite and `Scene::initPatches` derives scale and normal from it -- Scene.cpp:123-163 -- but a C ABI has to survive what it
is handed).  Asserted:

  * the call returns (no hang, no fault), every hostile patch comes back either failed and untouched or with finite fields;
  * THE OTHER PATCHES OF THE BATCH ARE NOT AFFECTED: their results are byte-identical to the same patches refined without them;
  * the oracle (the reference's arithmetic on the same floats: NaN comparisons, log2 of 0, a zero normal ...) takes the SAME
    decision at the SAME stage for every one of them, and the one that survives (an image attached three times) is refined
    bit for bit like the oracle refines it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_hostile_patches_do_not_hang_and_do_not_disturb_their_neighbours(tiny_scene, tiny_seeds, gpu_scene, oracle_scene):
    from hpmvs_amd import api
    n = 192
    clean = api.Batch.from_seeds(tiny_seeds, np.arange(n))
    api.optimize_batch(gpu_scene, clean)
    assert clean.ok.sum() > n // 2

    mixed = api.Batch.from_seeds(tiny_seeds, np.arange(n))
    cam0 = np.asarray(tiny_scene.views[0].c, dtype=np.float32)
    nan, inf = np.float32(np.nan), np.float32(np.inf)
    hostile = {
        3: lambda b, k: b.center.__setitem__((k, 0), nan),
        11: lambda b, k: b.center.__setitem__((k, slice(0, 3)), inf),
        17: lambda b, k: b.normal.__setitem__((k, slice(0, 3)), 0.0),
        23: lambda b, k: b.normal.__setitem__((k, 1), nan),
        31: lambda b, k: b.scale.__setitem__(k, 0.0),
        37: lambda b, k: b.scale.__setitem__(k, -1.0),
        41: lambda b, k: b.scale.__setitem__(k, np.float32(1e30)),
        47: lambda b, k: b.scale.__setitem__(k, nan),
        53: lambda b, k: b.center.__setitem__((k, slice(0, 3)), cam0),          # the patch sits in a camera centre
        59: lambda b, k: b.center.__setitem__((k, slice(0, 3)), np.float32(1e20)),
        61: lambda b, k: b.n_images.__setitem__(k, 0),
        67: lambda b, k: b.images.__setitem__((k, slice(0, 3)), b.images[k, 0]),  # the same image three times
        71: lambda b, k: b.normal.__setitem__((k, slice(0, 3)), np.float32(1e-30)),
        73: lambda b, k: b.center.__setitem__((k, 2), -inf),
    }
    for k, f in hostile.items():
        f(mixed, k)
    before = {f: getattr(mixed, f).copy() for f in ("center", "normal", "n_images", "images")}
    api.optimize_batch(gpu_scene, mixed)   # must return
    bad = sorted(hostile)
    good = np.array([k for k in range(n) if k not in hostile])
    # the neighbours: exactly what they are without the hostile patches around them
    for f in ("ok", "center", "normal", "n_images", "images", "color", "fmin", "x", "nevals", "stage", "ngrabs"):
        assert np.array_equal(getattr(mixed, f)[good], getattr(clean, f)[good]), f
    # the hostile ones: failed and untouched (bit patterns: NaN != NaN), or refined to finite fields
    report = {}
    for k in bad:
        if mixed.ok[k]:
            assert np.all(np.isfinite(mixed.center[k])) and np.all(np.isfinite(mixed.normal[k])) and np.all(np.isfinite(mixed.color[k])), k
            assert 3 <= mixed.n_images[k] <= mixed.max_images
        else:
            assert mixed.center[k].tobytes() == before["center"][k].tobytes() and mixed.normal[k].tobytes() == before["normal"][k].tobytes(), k
            assert mixed.n_images[k] == before["n_images"][k] and np.array_equal(mixed.images[k], before["images"][k]), k
            assert mixed.stage[k] > 0
        assert 0 <= mixed.nevals[k] <= 1000
        report[k] = (int(mixed.ok[k]), int(mixed.stage[k]), int(mixed.nevals[k]))
    print("hostile patches (ok, failing stage, evaluations):", report)
    # the oracle on the same hostile floats
    from hpmvs_amd import synth
    from oracle import oracle as orc
    hs = synth.Seeds(center=before["center"], normal=before["normal"], scale=mixed.scale.copy(), n_images=before["n_images"],
                     images=before["images"], truth=np.zeros((n, 3)))
    P = orc.patches_from_seeds(hs, bad)
    orc.optimize_batch(oracle_scene, P, n_threads=4)
    for p, k in zip(P, bad):
        assert p.stage == mixed.stage[k] and p.nevals == mixed.nevals[k], (k, p.stage, int(mixed.stage[k]))
        if p.stage == 0:
            assert np.array_equal(np.array(p.center[:], dtype=np.float32), mixed.center[k]) and np.array_equal(np.array(p.normal[:], dtype=np.float32), mixed.normal[k])
            assert list(p.images[:p.n_images]) == list(mixed.images[k, :mixed.n_images[k]])
