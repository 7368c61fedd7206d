"""Image lists beyond the batch kernel's 64-id rows (reference: `std::vector<int> pImages_`,
include/hpmvs/PatchOptimizer.h:106, push_back at src/hpmvs/PatchOptimizer.cpp:254 -- unbounded).

A 200-view scene at START_LEVEL 1: addImages attaches far more than 64 views to many seeds before the first
filterImagesNCC cuts the list down.  Such a patch ends in the batch kernel (ok = 2) and is redone by the wide kernel behind
it (kernel_wide.hip, lists up to HPMVS_MAX_IMAGES = 256).  What must hold: NO stage-100 patch, and EVERY patch -- whichever
kernel produced it -- equals the oracle (genuine NLopt; ORC_MAX_IMAGES = 256) in stage, refined fields, list, evaluation
count."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _compare(tag, scene, gpu, seeds, min_wide):
    from hpmvs_amd import api
    from oracle import oracle as orc
    assert orc.optimizer_available(orc.OPT_REF)
    b = api.Batch.from_seeds(seeds)
    api.optimize_batch(gpu, b)
    n_wide = gpu.last_wide_patches()
    osc = orc.OracleScene(scene)
    P = orc.patches_from_seeds(seeds)
    orc.optimize_batch(osc, P, which=orc.OPT_REF, n_threads=os.cpu_count() or 8)
    st = np.array([p.stage for p in P])
    assert int((b.stage == 100).sum()) == 0 and int((st == 100).sum()) == 0, (tag, int((b.stage == 100).sum()), int((st == 100).sum()))
    assert n_wide >= min_wide, (tag, "the scene does not exercise the wide kernel", n_wide)
    assert np.array_equal(st, b.stage), (tag, np.nonzero(st != b.stage)[0][:10], st[st != b.stage][:10], b.stage[st != b.stage][:10])
    ok = np.nonzero(b.ok)[0]
    assert len(ok) >= 20, (tag, len(ok))
    differ = []
    for k in ok:
        same = (np.array_equal(np.array(P[k].center[:], dtype=np.float32), b.center[k]) and
                np.array_equal(np.array(P[k].normal[:], dtype=np.float32), b.normal[k]) and
                np.array_equal(np.array(P[k].color[:], dtype=np.float32), b.color[k]) and
                P[k].nevals == b.nevals[k] and P[k].fmin == b.fmin[k] and
                list(P[k].images[:P[k].n_images]) == list(b.images[k, :b.n_images[k]]))
        if not same:
            differ.append(int(k))
    # the one known source of differing bits: glibc's acos is not correctly rounded, the device's is
    from helpers import equals_gpu_with_correctly_rounded_acos
    for k in differ:
        assert equals_gpu_with_correctly_rounded_acos(osc, orc.default_options(), seeds, k, b), (tag, k)
    rep = dict(tag=tag, patches=int(b.n), refined=int(len(ok)), redone_by_the_wide_kernel=int(n_wide), stage_100=0,
               longest_refined_list=int(b.n_images[ok].max()), differing_patches_all_of_the_acos_class=len(differ),
               launch_ms=float(gpu.last_kernel_ms()), stage_histogram={int(s): int(c) for s, c in zip(*np.unique(b.stage, return_counts=True))})
    print(rep)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/wide_lists.jsonl", "a") as fh:
        fh.write(json.dumps(rep) + "\n")
    return b, P


def test_200_view_scene_at_start_level_1_no_overflow_and_every_patch_equals_the_oracle():
    import torch
    from hpmvs_amd import api, synth
    scene = synth.make_scene(200, 960, 540, n_waves=24, device=torch.device("cuda", 0))
    gpu = api.Scene(scene, device=0)
    try:
        seeds = synth.make_seeds(scene, 1500, start_level=1, max_images=256, seed=synth.SEED + 21)
        _compare("200v_960x540_start_level_1", scene, gpu, seeds, min_wide=50)
    finally:
        gpu.close()


def test_seed_lists_longer_than_64_ids_and_rows_of_every_width():
    """Seeds that ARRIVE with more than 64 attached images (a dense NVM point), rows of 96 / 200 / 256 ids; and a batch whose
    rows are too narrow for a refined list still fails that patch with stage 100, untouched."""
    import torch
    from hpmvs_amd import api, synth
    scene = synth.make_scene(120, 640, 360, n_waves=16, device=torch.device("cuda", 0))
    gpu = api.Scene(scene, device=0)
    try:
        seeds = synth.make_seeds(scene, 300, start_level=1, max_images=256, max_seed_images=110, seed=synth.SEED + 22)
        assert seeds.n_images.max() > 64
        b, P = _compare("120v_640x360_long_seed_lists", scene, gpu, seeds, min_wide=30)
        for cols in (200, 128):
            if seeds.n_images.max() > cols:
                continue
            s2 = synth.Seeds(center=seeds.center, normal=seeds.normal, scale=seeds.scale, n_images=seeds.n_images,
                             images=seeds.images[:, :cols].copy(), truth=seeds.truth)
            b2 = api.Batch.from_seeds(s2)
            api.optimize_batch(gpu, b2)
            fits = b.n_images <= cols
            assert np.array_equal(b2.stage[fits], b.stage[fits]) and np.array_equal(b2.center[fits], b.center[fits])
            assert (b2.stage[~fits & (b.ok != 0)] == 100).all()
        # the unchanged callers' path (a batch of one through the open batch) gives the same patch as the batch did
        wide_ones = [k for k in range(b.n) if b.ok[k] and seeds.n_images[k] <= 64][:3]
        for k in wide_ones:
            one = api.Batch.from_seeds(seeds, np.array([k]))
            one64 = api.Batch(one.center, one.normal, one.scale, one.n_images, one.images[:, :64])
            api.optimize_batch(gpu, one64)
            if b.n_images[k] <= 64:
                assert one64.ok[0] == 1 and np.array_equal(one64.center[0], b.center[k]) and one64.nevals[0] == b.nevals[k]
    finally:
        gpu.close()


def test_unchanged_single_patch_callers_on_a_dense_scene(tmp_path):
    """The reference's calling pattern -- optimize(patch), one patch per call (CellProcessor.cpp:129,256) -- on a scene where the
    lists outgrow 64 ids: a call goes through the open batch (64-id records), comes back with stage 100, is re-run through a batch
    launch of its own (capi.hip) and, if its REFINED list is longer than 64 ids, once more with rows as wide as the scene's views
    (hpmvs_host.cpp).  Every call must give the patch that one optimizeBatch() over all of them gives (rows of min(views, 256))."""
    import json
    import subprocess
    import sys
    from hpmvs_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests"))
    from test_gpu_cpp_interface import _dump_scene
    exe = os.path.join(root, "tests", "native", "bench_callers")
    scene = synth.make_scene(120, 640, 360, n_waves=16)
    n = 160
    seeds = synth.make_seeds(scene, n, start_level=1, max_images=256, seed=synth.SEED + 22)
    dump = tmp_path / "dense.bin"
    _dump_scene(dump, scene, seeds, n)
    env = dict(os.environ, HPMVS_SERVICE="1", OMP_WAIT_POLICY="passive")
    r = subprocess.run([exe, str(dump), "8"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    row = json.loads(r.stdout.strip().splitlines()[-1])
    print(row)
    assert row["mismatches"] == 0 and row["refined"] > n // 2, row


def test_hostile_and_overflowing_patches_in_the_wide_kernel():
    """What the batch kernel survives (tests/test_gpu_hostile_patches.py) the wide kernel must survive too, with the oracle's decision:
    non-finite and degenerate patches whose lists are long enough to be redone by it, an image attached several times, and lists
    that outgrow even 256 ids (a 600-view scene: both sides end at stage 100, patch untouched).  The neighbours are unaffected."""
    import torch
    from hpmvs_amd import api, synth
    from oracle import oracle as orc
    scene = synth.make_scene(600, 160, 90, n_waves=12, device=torch.device("cuda", 0))
    gpu = api.Scene(scene, device=0)
    try:
        n = 240
        seeds = synth.make_seeds(scene, n, start_level=0, max_images=256, max_seed_images=100, seed=synth.SEED + 23)
        assert seeds.n_images.max() > 64
        clean = api.Batch.from_seeds(seeds)
        api.optimize_batch(gpu, clean)
        n_wide_clean = gpu.last_wide_patches()
        mixed = api.Batch.from_seeds(seeds)
        nan, inf = np.float32(np.nan), np.float32(np.inf)
        long_ones = [int(k) for k in np.nonzero(seeds.n_images > 70)[0][:8]]
        assert len(long_ones) == 8
        hostile = {
            long_ones[0]: lambda b, k: b.center.__setitem__((k, 0), nan),
            long_ones[1]: lambda b, k: b.normal.__setitem__((k, slice(0, 3)), 0.0),
            long_ones[2]: lambda b, k: b.scale.__setitem__(k, 0.0),
            long_ones[3]: lambda b, k: b.scale.__setitem__(k, nan),
            long_ones[4]: lambda b, k: b.center.__setitem__((k, slice(0, 3)), inf),
            long_ones[5]: lambda b, k: b.images.__setitem__((k, slice(0, 5)), b.images[k, 0]),   # the same image five times
            long_ones[6]: lambda b, k: b.scale.__setitem__(k, np.float32(1e30)),
            long_ones[7]: lambda b, k: b.normal.__setitem__((k, 1), nan),
        }
        for k, f in hostile.items():
            f(mixed, k)
        before = {f: getattr(mixed, f).copy() for f in ("center", "normal", "scale", "n_images", "images")}
        api.optimize_batch(gpu, mixed)    # must return
        assert gpu.last_wide_patches() >= 8
        good = np.array([k for k in range(n) if k not in hostile])
        for f in ("ok", "center", "normal", "n_images", "images", "color", "fmin", "nevals", "stage", "ngrabs"):
            assert np.array_equal(getattr(mixed, f)[good], getattr(clean, f)[good]), f
        osc = orc.OracleScene(scene)
        hs = synth.Seeds(center=before["center"], normal=before["normal"], scale=before["scale"], n_images=before["n_images"],
                         images=before["images"], truth=np.zeros((n, 3)))
        P = orc.patches_from_seeds(hs)
        orc.optimize_batch(osc, P, which=orc.OPT_REF, n_threads=os.cpu_count() or 8)
        st = np.array([p.stage for p in P])
        assert np.array_equal(st, mixed.stage), (np.nonzero(st != mixed.stage)[0][:10], st[st != mixed.stage][:10], mixed.stage[st != mixed.stage][:10])
        for k in range(n):
            if mixed.ok[k]:
                assert np.all(np.isfinite(mixed.center[k])) and np.all(np.isfinite(mixed.normal[k]))
                assert np.array_equal(np.array(P[k].center[:], dtype=np.float32), mixed.center[k]), k
                assert list(P[k].images[:P[k].n_images]) == list(mixed.images[k, :mixed.n_images[k]]) and P[k].nevals == mixed.nevals[k], k
            else:
                assert mixed.center[k].tobytes() == before["center"][k].tobytes() and np.array_equal(mixed.images[k], before["images"][k]), k
        assert (mixed.stage == 100).sum() >= 5, "the scene does not push any list beyond 256 ids"   # (both sides: st == mixed.stage above)
        rep = dict(tag="600v_160x90_start_level_0_hostile", patches=n, redone_by_the_wide_kernel_clean=int(n_wide_clean), refined=int(mixed.ok.sum()),
                   stage_histogram={int(s): int(c) for s, c in zip(*np.unique(mixed.stage, return_counts=True))},
                   hostile={int(k): (int(mixed.ok[k]), int(mixed.stage[k])) for k in hostile})
        print(rep)
        with open("gpurun_out/wide_lists.jsonl", "a") as fh:
            fh.write(json.dumps(rep) + "\n")
    finally:
        gpu.close()
