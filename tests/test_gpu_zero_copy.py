"""Host-pointer calls on pinned host memory use the arrays in place (include/hpmvs_amd.h: "Host-pointer calls and pinned
memory"; capi.hip: mapped_host).  The reference keeps its patches in host containers (Scene.cpp:94-96); an integrator who
allocates the batch arrays with hpmvs_host_alloc / hipHostMalloc saves the staging copies.  Same results as the staged call,
array by array, also when only some arrays are pinned."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ("center", "normal", "n_images", "images", "ok", "color", "ncc", "fmin", "x", "result", "nevals", "stage", "ngrabs")


def _equal(a, b):
    for k in FIELDS:
        assert np.array_equal(getattr(a, k), getattr(b, k)), k


def test_pinned_batch_is_refined_in_place_with_identical_results(tiny_scene, gpu_scene):
    from hpmvs_amd import api, synth
    seeds = synth.make_seeds(tiny_scene, 20000, start_level=2, max_images=16, seed=synth.SEED + 77)
    staged = api.Batch.from_seeds(seeds)
    staged.color[:] = 7.0   # a staged call returns 0 for what the kernel leaves untouched: so must the in-place call
    api.optimize_batch(gpu_scene, staged)
    s_bytes, p_bytes = gpu_scene.last_staging()
    assert s_bytes > 0 and p_bytes == 0
    assert 0 < staged.ok.sum() < staged.n   # both kinds of patches

    pinned = api.Batch.from_seeds(seeds).pin()
    pinned.color[:] = 7.0
    pinned.ncc[:] = 7.0
    api.optimize_batch(gpu_scene, pinned)
    s_bytes, p_bytes = gpu_scene.last_staging()
    assert s_bytes == 0 and p_bytes == sum(getattr(pinned, k).nbytes for k in api.Batch.FIELDS)
    _equal(staged, pinned)
    assert np.all(pinned.color[pinned.ok == 0] == 0.0) and np.all(pinned.ncc[pinned.ok == 0] == 0.0)

    # some arrays pinned, some pageable: array by array
    mixed = api.Batch.from_seeds(seeds).pin()
    mixed.center = mixed.center.copy()
    mixed.images = mixed.images.copy()
    mixed.fmin = mixed.fmin.copy()
    api.optimize_batch(gpu_scene, mixed)
    s_bytes, p_bytes = gpu_scene.last_staging()
    assert s_bytes == mixed.center.nbytes + mixed.images.nbytes + mixed.fmin.nbytes and p_bytes > 0
    _equal(staged, mixed)


def test_zero_copy_can_be_switched_off(tiny_scene, gpu_scene):
    """HPMVS_ZERO_COPY is read once per process: a child process with it set to 0 stages pinned arrays like pageable ones."""
    import os
    import subprocess
    import sys
    code = ("import numpy as np\n"
            "from hpmvs_amd import api, synth\n"
            "sc = synth.make_scene(3, 640, 480, n_waves=8)\n"
            "g = api.Scene(sc)\n"
            "b = api.Batch.from_seeds(synth.make_seeds(sc, 20000, start_level=2, max_images=16)).pin()\n"
            "api.optimize_batch(g, b)\n"
            "print('STAGING', *g.last_staging())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HPMVS_ZERO_COPY="0", PYTHONPATH=root), cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("STAGING")][-1].split()
    assert int(line[1]) > 0 and int(line[2]) == 0
