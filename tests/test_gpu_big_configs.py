"""BASELINE.json configs[1] (12 x 1080p, 1e5 seeds) and configs[2] (50 x 4K, 1e6 seeds: the configuration the
headline metric is quoted on) through the C ABI at FULL size.

Two kinds of checks per configuration:
  * sampled oracle parity: the first 8 192 seeds against the CPU oracle running the reference's GENUINE NLopt
    BOBYQA (oracle/_ref, asserted -- the port of the product's own optimiser is never used here).  Tolerance as
    in test_gpu_optimize.py: decisions + failing stage identical, >= 99.9 % of the jointly refined patches
    bit-identical in every output field -- the floor is 99.9 % (measured: 100 % C1, 99.997 % C2), mean |d ncc| <= 1e-5;
  * size-independent properties over the whole batch: a failed patch is untouched (PatchOptimizer.cpp:86-87),
    ncc_ == 1.4f and w components 1 / 0 on success, image ids inside the scene and unique, at least
    MIN_IMAGES_PER_PATCH images, two launches give identical bytes (the work queue hands patches to wavefronts
    in a different order every time), and the sample's results do not depend on the batch they were part of.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SAMPLE = 8192


def _device_scene(views, w, h):
    import torch
    from hpmvs_amd import api, synth
    if api.device_count() < 1:
        pytest.fail("no HIP device: -m gpu tests need the MI355X box (no CPU fallback exists)")
    scene = synth.make_scene(views, w, h, n_waves=24, device=torch.device("cuda", 0))
    return scene, api.Scene(scene, device=0)


def _run_config(views, w, h, n, tag):
    from hpmvs_amd import api, synth
    from oracle import oracle as orc
    assert orc.optimizer_available(orc.OPT_REF), "oracle/_ref (genuine NLopt BOBYQA) did not travel to this box"
    scene, gpu = _device_scene(views, w, h)
    seeds = synth.make_seeds(scene, n, start_level=4, max_images=32)
    batch = api.Batch.from_seeds(seeds)
    before = (batch.center.copy(), batch.normal.copy(), batch.n_images.copy(), batch.images.copy())
    api.optimize_batch(gpu, batch)
    ok = batch.ok.astype(bool)
    bad = ~ok
    assert ok.sum() > 0.3 * n, (tag, ok.mean())

    # ---- properties over the FULL batch
    assert np.array_equal(batch.center[bad], before[0][bad]) and np.array_equal(batch.normal[bad], before[1][bad])
    assert np.array_equal(batch.n_images[bad], before[2][bad]) and np.array_equal(batch.images[bad], before[3][bad])
    assert np.all(batch.ncc[ok] == np.float32(1.4))
    assert np.all(batch.center[ok, 3] == 1.0) and np.all(batch.normal[ok, 3] == 0.0)
    nim = batch.n_images[ok]
    assert nim.min() >= 3 and nim.max() <= batch.max_images
    img = batch.images[ok]
    live = np.arange(batch.max_images)[None, :] < nim[:, None]
    assert img[live].min() >= 0 and img[live].max() < views
    srt = np.sort(np.where(live, img, -1 - np.arange(batch.max_images)[None, :]), axis=1)
    assert np.all(srt[:, 1:] != srt[:, :-1]), "an image id is attached twice"
    assert np.all(np.isfinite(batch.center[ok])) and np.all(np.isfinite(batch.color[ok]))
    nn = np.linalg.norm(batch.normal[ok, :3].astype(np.float64), axis=1)
    assert np.abs(nn - 1.0).max() < 1e-5
    assert np.all((batch.stage[ok] == 0)) and np.all(batch.stage[bad] > 0)
    assert np.all((batch.result[ok] >= 1) & (batch.result[ok] <= 4))  # PatchOptimizer.cpp:367-368
    assert batch.nevals[ok].min() >= 8 and batch.nevals.max() <= 1000

    # ---- determinism: a second launch of the same batch, and the sample alone, give the same bytes
    again = api.Batch.from_seeds(seeds)
    api.optimize_batch(gpu, again)
    for f in ("ok", "center", "normal", "n_images", "images", "color", "fmin", "x", "nevals", "stage", "ngrabs"):
        assert np.array_equal(getattr(again, f), getattr(batch, f)), (tag, f)
    m = min(SAMPLE, n)
    alone = api.Batch.from_seeds(seeds, np.arange(m))
    api.optimize_batch(gpu, alone)
    for f in ("ok", "center", "normal", "n_images", "images", "color", "fmin", "nevals", "stage"):
        assert np.array_equal(getattr(alone, f), getattr(batch, f)[:m]), (tag, f)

    # ---- sampled oracle parity (genuine NLopt)
    osc = orc.OracleScene(scene)
    P = orc.patches_from_seeds(seeds, np.arange(m))
    import os
    orc.optimize_batch(osc, P, which=orc.OPT_REF, n_threads=os.cpu_count() or 8)
    ok_cpu = np.array([p.stage == 0 for p in P])
    st_cpu = np.array([p.stage for p in P])
    assert np.array_equal(ok_cpu, ok[:m]), tag
    assert np.array_equal(st_cpu, batch.stage[:m]), tag
    both = np.nonzero(ok_cpu)[0]
    unrob = lambda f: f / (1.0 - 3.0 * f)
    ident = np.array([
        np.array_equal(np.array(P[k].center[:], dtype=np.float32), batch.center[k]) and
        np.array_equal(np.array(P[k].normal[:], dtype=np.float32), batch.normal[k]) and
        np.array_equal(np.array(P[k].color[:], dtype=np.float32), batch.color[k]) and
        P[k].nevals == batch.nevals[k] and
        list(P[k].images[:P[k].n_images]) == list(batch.images[k, :batch.n_images[k]]) for k in both])
    dncc = np.array([abs(unrob(P[k].fmin) - unrob(batch.fmin[k])) for k in both])
    grabs_same = float(np.mean([P[k].ngrabs == batch.ngrabs[k] for k in both]))
    report = dict(tag=tag, n=n, sample=m, grab_count_agreement=grabs_same, ok_fraction=float(ok.mean()), ok_sample=int(ok_cpu.sum()),
                  bit_identical_patch=float(ident.mean()), mean_dncc=float(dncc.mean()),
                  kernel_ms=gpu.last_kernel_ms())
    print(report)
    import json
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_reports.jsonl", "a") as fh:
        fh.write(json.dumps(report) + "\n")
    assert ident.mean() >= 0.999, report   # measured: 1.0 (C1, C4), 0.99997 (C2: one glibc-acos patch)
    # ... and whatever differs does so for the one known reason: with the oracle's acos() correctly rounded (what the device
    # computes) the oracle gives the GPU's result for that patch (helpers.equals_gpu_with_correctly_rounded_acos)
    from helpers import equals_gpu_with_correctly_rounded_acos
    for k in both[~ident]:
        assert equals_gpu_with_correctly_rounded_acos(osc, orc.default_options(), seeds, int(k), batch), (report, int(k))
    assert dncc.mean() <= 1e-5, report
    gpu.close()


def test_c1_12_views_1080p_1e5_seeds():
    _run_config(12, 1920, 1080, 100_000, "c1_12v_1080p_1e5")


def test_c2_50_views_4k_1e6_seeds():
    _run_config(50, 3840, 2160, 1_000_000, "c2_50v_4k_1e6")


def test_c4_200_views_4k_2e5_seeds():
    """BASELINE.json configs[4]'s scene at full size -- 200 x 3840x2160 views, 6.6 GB of pyramids rendered on the
    device, covisibility lists of up to 199 views (addImages walks them in chunks of 64 lanes), 16-bit image ids
    -- with 2e5 of its 1e7 seeds so that the test stays under a minute.  Same oracle sample and property block."""
    _run_config(200, 3840, 2160, 200_000, "c4_200v_4k_2e5")


def test_c4_200_views_4k_1e7_seeds():
    """BASELINE.json configs[4] at its STATED seed count on one GPU: 200 x 3840x2160 views (6.6 GB of pyramids) and
    1e7 seed patches in ONE hpmvs_optimize_batch call (VERDICT r4 #1a; the 2e5-seed test above keeps the full
    property block).  The seeds are built on the device (synth.make_seeds(device=...): 200 visibility passes over 1e7
    points are minutes of numpy on the host).  Checked here, sized to stay under two minutes:
      * two launches of the whole batch give identical bytes in every output array;
      * a failed patch is untouched, ncc_ == 1.4f / w components on success, stage / result codes consistent -- over
        the FULL batch; image-id range and uniqueness on a strided sample of 1 in 16;
      * the first 8 192 seeds against the oracle with the reference's genuine NLopt (asserted), bit for bit;
      * those 8 192 refined ALONE give the bytes they got inside the 1e7 batch."""
    import os
    import torch
    from hpmvs_amd import api, synth
    from oracle import oracle as orc
    assert orc.optimizer_available(orc.OPT_REF), "oracle/_ref (genuine NLopt BOBYQA) did not travel to this box"
    views, n, m = 200, 10_000_000, SAMPLE
    scene, gpu = _device_scene(views, 3840, 2160)
    seeds = synth.make_seeds(scene, n, start_level=4, max_images=32, device=torch.device("cuda", 0))
    assert len(seeds.scale) == n
    batch = api.Batch.from_seeds(seeds)
    api.optimize_batch(gpu, batch)
    kernel_ms = gpu.last_kernel_ms()
    ok = batch.ok.astype(bool)
    bad = ~ok
    assert 0.2 * n < ok.sum() < n, ok.mean()
    # ---- the whole batch
    assert np.array_equal(batch.center[bad], seeds.center[bad]) and np.array_equal(batch.normal[bad], seeds.normal[bad])
    assert np.array_equal(batch.n_images[bad], seeds.n_images[bad]) and np.array_equal(batch.images[bad], seeds.images[bad])
    assert np.all(batch.ncc[ok] == np.float32(1.4))
    assert np.all(batch.center[ok, 3] == 1.0) and np.all(batch.normal[ok, 3] == 0.0)
    assert np.all(batch.stage[ok] == 0) and np.all(batch.stage[bad] > 0)
    assert np.all((batch.result[ok] >= 1) & (batch.result[ok] <= 4))
    assert batch.nevals[ok].min() >= 8 and batch.nevals.max() <= 1000
    assert batch.n_images[ok].min() >= 3 and batch.n_images[ok].max() <= batch.max_images
    assert np.all(np.isfinite(batch.center[ok])) and np.all(np.isfinite(batch.color[ok]))
    # ---- a strided sample: ids inside the scene, no id attached twice
    sl = np.nonzero(ok[::16])[0] * 16
    nim, img = batch.n_images[sl], batch.images[sl]
    live = np.arange(batch.max_images)[None, :] < nim[:, None]
    assert img[live].min() >= 0 and img[live].max() < views
    srt = np.sort(np.where(live, img, -1 - np.arange(batch.max_images)[None, :]), axis=1)
    assert np.all(srt[:, 1:] != srt[:, :-1]), "an image id is attached twice"
    # ---- determinism: a second launch of all 1e7, and the sample alone
    again = api.Batch.from_seeds(seeds)
    api.optimize_batch(gpu, again)
    for f in ("ok", "center", "normal", "n_images", "images", "color", "fmin", "x", "nevals", "stage", "ngrabs"):
        assert np.array_equal(getattr(again, f), getattr(batch, f)), f
    del again
    alone = api.Batch.from_seeds(seeds, np.arange(m))
    api.optimize_batch(gpu, alone)
    for f in ("ok", "center", "normal", "n_images", "images", "color", "fmin", "nevals", "stage"):
        assert np.array_equal(getattr(alone, f), getattr(batch, f)[:m]), f
    # ---- oracle sample (genuine NLopt)
    osc = orc.OracleScene(scene)
    P = orc.patches_from_seeds(seeds, np.arange(m))
    orc.optimize_batch(osc, P, which=orc.OPT_REF, n_threads=os.cpu_count() or 8)
    st_cpu = np.array([p.stage for p in P])
    assert np.array_equal(st_cpu, batch.stage[:m])
    both = np.nonzero(st_cpu == 0)[0]
    ident = np.array([
        np.array_equal(np.array(P[k].center[:], dtype=np.float32), batch.center[k]) and
        np.array_equal(np.array(P[k].normal[:], dtype=np.float32), batch.normal[k]) and
        np.array_equal(np.array(P[k].color[:], dtype=np.float32), batch.color[k]) and
        P[k].nevals == batch.nevals[k] and
        list(P[k].images[:P[k].n_images]) == list(batch.images[k, :batch.n_images[k]]) for k in both])
    unrob = lambda f: f / (1.0 - 3.0 * f)
    dncc = np.array([abs(unrob(P[k].fmin) - unrob(batch.fmin[k])) for k in both])
    report = dict(tag="c4_200v_4k_1e7", n=n, sample=m, ok_fraction=float(ok.mean()), ok_sample=int(len(both)),
                  bit_identical_patch=float(ident.mean()), mean_dncc=float(dncc.mean()), kernel_ms=kernel_ms,
                  patches_per_s=n / (kernel_ms * 1e-3), mean_grabs_per_patch=float(batch.ngrabs.mean()))
    print(report)
    import json
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_reports.jsonl", "a") as fh:
        fh.write(json.dumps(report) + "\n")
    assert ident.mean() >= 0.999 and dncc.mean() <= 1e-5, report
    from helpers import equals_gpu_with_correctly_rounded_acos
    for k in both[~ident]:  # (none measured) a difference is allowed for the one known reason only
        assert equals_gpu_with_correctly_rounded_acos(osc, orc.default_options(), seeds, int(k), batch), (report, int(k))
    gpu.close()


def test_c2_every_patch_against_the_oracle():
    """BASELINE configs[2] without sampling: all 1e6 patches through the GPU and through the oracle (genuine NLopt, all granted
    cores, ~17 s), compared bit for bit -- the failing stage of every patch; centre, normal, colour, image list and evaluation
    count of every refined one (tools/full_parity.py).  Measured: 49 of 968 000 refined patches differ.  The claim of DESIGN
    section 5 is tested for EVERY one of them: run again with the oracle's acos() correctly rounded (libquadmath; what the device's
    cr_acos.hpp returns) the oracle equals the GPU -- nothing differs for any other reason, and no list outgrows the 32-id rows."""
    import sys
    from hpmvs_amd import api
    from oracle import oracle as orc
    if api.device_count() < 1:
        pytest.fail("no HIP device: -m gpu tests need the MI355X box (no CPU fallback exists)")
    assert orc.optimizer_available(orc.OPT_REF), "oracle/_ref (genuine NLopt BOBYQA) did not travel to this box"
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import full_parity
    rec = full_parity.run(50, 3840, 2160, 1_000_000)
    brief = {k: v for k, v in rec.items() if k != "differing_patches"}
    print(brief)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/full_parity_c2.json", "w") as fh:
        import json
        fh.write(json.dumps(rec) + "\n")
    assert rec["refined_by_both"] == rec["refined_gpu"] > 900_000, brief
    assert rec["differing"] <= 100, brief                      # measured: 49 (0.005 %)
    assert rec["unexplained"] == 0, (brief, rec["differing_patches"])
    assert rec["attribution"]["equal_with_correctly_rounded_acos"] == rec["differing"], brief
