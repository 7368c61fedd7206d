"""GPU parity of the full refinement (hpmvs_optimize_batch) against the CPU oracle, which runs the
restated photometric path around the reference's GENUINE NLopt BOBYQA (oracle/_ref) when that
library travelled, else the goldens it produced.

Stated tolerance.  Every float32 operation of the objective and every float64 operation of BOBYQA
is the one the CPU executes, so the whole refinement is expected to be BIT-IDENTICAL: same
success/failure and failing stage, same evaluation count, same centre / normal / image list /
colour.  The only arithmetic that is not reproduced operation-for-operation is the C library's
double cos()/log2() (used once per patch for cosb and the level tests); asinf IS reproduced and the
float64 acos of the start point is rounded to nearest on the device, which is what glibc returns for
all but ~0.06 % of arguments (photometric.hpp, cr_acos.hpp).  A last-bit difference there perturbs the
float64 start point by ~1e-16 and BOBYQA can amplify that (it moves 7 % of patches out of their basin
for a 1e-9 relative perturbation of f, tests/sensitivity_probe.py).  The tests therefore require
  * decisions + failing stage identical            : 100 %
  * bit-identical refined patch (all float fields) : >= 99 %  (measured 100 %)
  * mean |d ncc| over all jointly successful patches <= 1e-5  (SURVEY 8d proposed 5e-3)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def unrobust(f):
    return f / (1.0 - 3.0 * f)


def compare(seeds, oracle_scene, gpu_scene, tag):
    from hpmvs_amd import api
    from oracle import oracle as orc
    batch = api.Batch.from_seeds(seeds)
    api.optimize_batch(gpu_scene, batch)
    P = orc.patches_from_seeds(seeds)
    orc.optimize_batch(oracle_scene, P, which=orc.OPT_REF, n_threads=8)
    n = batch.n
    ok_cpu = np.array([P[k].stage == 0 for k in range(n)])
    ok_gpu = batch.ok.astype(bool)
    stage_cpu = np.array([P[k].stage for k in range(n)])
    agree = (ok_cpu == ok_gpu) & (stage_cpu == batch.stage)
    both = np.nonzero(ok_cpu & ok_gpu)[0]
    assert ok_cpu.sum() >= 20, f"{tag}: scene too hard for the oracle ({ok_cpu.sum()} successes)"
    dn, dd, same_imgs, dncc, dcol = [], [], [], [], []
    for k in both:
        p = P[k]
        cc = np.array(p.center[:3]); cg = batch.center[k, :3]
        nc = np.array(p.normal[:3]); ng = batch.normal[k, :3]
        dd.append(np.linalg.norm(cc - cg) / float(seeds.scale[k]))
        cosang = np.clip(np.dot(nc, ng) / (np.linalg.norm(nc) * np.linalg.norm(ng)), -1, 1)
        dn.append(np.degrees(np.arccos(cosang)))
        same_imgs.append(p.n_images == batch.n_images[k] and list(p.images[:p.n_images]) == list(batch.images[k, :p.n_images]))
        dncc.append(abs(unrobust(p.fmin) - unrobust(batch.fmin[k])))
        dcol.append(np.abs(np.array(p.color[:]) - batch.color[k]).max())
    dd, dn = np.array(dd), np.array(dn)
    basin = (dd < 0.25) & (dn < 2.0)
    report = dict(tag=tag, n=n, ok_cpu=int(ok_cpu.sum()), ok_gpu=int(ok_gpu.sum()), agree=float(agree.mean()),
                  same_basin=float(basin.mean()), mean_dncc=float(np.mean(dncc)), med_dd=float(np.median(dd)),
                  p95_dd=float(np.percentile(dd, 95)), med_dn=float(np.median(dn)), same_imgs=float(np.mean(same_imgs)),
                  identical_x=float(np.mean([np.array_equal(np.array(P[k].x[:]), batch.x[k]) for k in both])),
                  bit_identical_patch=float(np.mean([
                      np.array_equal(np.array(P[k].center[:], dtype=np.float32), batch.center[k]) and
                      np.array_equal(np.array(P[k].normal[:], dtype=np.float32), batch.normal[k]) and
                      np.array_equal(np.array(P[k].color[:], dtype=np.float32), batch.color[k]) and
                      P[k].nevals == batch.nevals[k] and
                      list(P[k].images[:P[k].n_images]) == list(batch.images[k, :batch.n_images[k]]) for k in both])),
                  nevals_cpu=float(np.mean([P[k].nevals for k in both])), nevals_gpu=float(batch.nevals[both].mean()),
                  max_dcol_same=float(np.max(np.array(dcol)[np.array(same_imgs)])) if any(same_imgs) else -1.0)
    print(report)
    import json, os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_reports.jsonl", "a") as fh:
        fh.write(json.dumps(report) + "\n")
    assert agree.mean() == 1.0, report
    assert report["bit_identical_patch"] >= 0.999, report  # measured: 1.0 on every scene of this file
    # The optimiser variables x[3] (float64, exported by the ABI).  x[0] (depth) and x[2] (asin of a float32, the
    # glibc algorithm restated on the device) must be identical.  x[1] starts from a float64 acos(): the device
    # rounds it to nearest (cr_acos.hpp), glibc 2.35 does not always (its acos is < 1 ulp, not correctly rounded),
    # and where the two differ the whole run carries an offset of a few ulp in x[1] that float32 centre / normal /
    # f never see (tools/x_mismatch_probe.py: 7 of 2850 patches, 1-4 ulp, every other output identical).  So: at
    # least 99 % identical, and for a patch whose outputs are bit-identical x may differ in x[1] only, by <= 16 ulp.
    assert report["identical_x"] >= 0.99 or report["identical_x"] >= report["bit_identical_patch"] - 0.02, report
    for k in both:
        xc, xg = np.array(P[k].x[:]), batch.x[k]
        same_out = (np.array_equal(np.array(P[k].center[:], dtype=np.float32), batch.center[k]) and
                    np.array_equal(np.array(P[k].normal[:], dtype=np.float32), batch.normal[k]) and P[k].nevals == batch.nevals[k])
        if same_out and not np.array_equal(xc, xg):
            assert xc[0] == xg[0] and xc[2] == xg[2], (tag, k, xc, xg)
            assert abs(xc[1] - xg[1]) <= 16 * np.spacing(abs(xc[1])), (tag, k, xc, xg)
    assert np.mean(dncc) <= 1e-5, report
    assert basin.mean() >= 0.999 and np.mean(same_imgs) >= 0.999, report
    # scale_3dx_ is returned unchanged and ncc_ is the reference's constant
    assert np.all(batch.ncc[ok_gpu] == np.float32(1.4))
    return report


def test_optimize_parity_tiny(tiny_seeds, oracle_scene, gpu_scene):
    compare(tiny_seeds, oracle_scene, gpu_scene, "tiny/start_level2")


def test_optimize_parity_level3(tiny_scene, oracle_scene, gpu_scene):
    from hpmvs_amd import synth
    seeds = synth.make_seeds(tiny_scene, 300, start_level=3, seed=synth.SEED + 7)
    compare(seeds, oracle_scene, gpu_scene, "tiny/start_level3")


def test_failures_leave_patch_untouched(tiny_scene, gpu_scene):
    """optimize() returns false without touching the patch (reference PatchOptimizer.cpp:86-87)."""
    from hpmvs_amd import api, synth
    seeds = synth.make_seeds(tiny_scene, 64, start_level=4)  # level-4 seeds cannot attach enough images here
    batch = api.Batch.from_seeds(seeds)
    before = (batch.center.copy(), batch.normal.copy(), batch.n_images.copy(), batch.images.copy())
    api.optimize_batch(gpu_scene, batch)
    bad = ~batch.ok.astype(bool)
    assert bad.sum() > 0
    assert np.array_equal(batch.center[bad], before[0][bad])
    assert np.array_equal(batch.normal[bad], before[1][bad])
    assert np.array_equal(batch.n_images[bad], before[2][bad])
    assert np.array_equal(batch.images[bad], before[3][bad])


def test_empty_and_ragged_batches(tiny_seeds, gpu_scene):
    from hpmvs_amd import api
    empty = api.Batch(np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32), np.zeros(0, np.float32),
                      np.zeros(0, np.int32), np.zeros((0, 8), np.int32))
    api.optimize_batch(gpu_scene, empty)
    # patches with 0 or 1 attached images fail at stage 1, a single patch batch works
    b = api.Batch.from_seeds(tiny_seeds, np.arange(1))
    b.n_images[:] = 0
    api.optimize_batch(gpu_scene, b)
    assert b.ok[0] == 0 and b.stage[0] == 1
    one = api.Batch.from_seeds(tiny_seeds, np.nonzero(tiny_seeds.n_images >= 3)[0][:1])
    api.optimize_batch(gpu_scene, one)
    assert one.stage[0] in (0, 2, 3, 4, 5, 6, 7, 8, 9)


def test_gpu_against_committed_goldens():
    """GPU path vs tests/golden (oracle + genuine NLopt, generated in the build container): does not
    depend on oracle/_ref having travelled to the GPU box."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import golden_scene
    from hpmvs_amd import api
    scene, seeds, exp = golden_scene()
    gpu = api.Scene(scene, device=0)
    batch = api.Batch.from_seeds(seeds)
    # objective probes (pins F12-F15 / A4-A10 without the optimiser)
    by_patch = {}
    for pr in exp["objective_probes"]:
        by_patch.setdefault(pr["patch"], []).append(pr)
    for j in range(4):
        idx = np.array(sorted(by_patch))
        x = np.array([by_patch[k][j]["x"] for k in idx])
        want = np.array([by_patch[k][j]["f"] for k in idx])
        got, _ = api.objective_batch(gpu, api.Batch.from_seeds(seeds, idx), x)
        assert np.array_equal(got, want)
    api.optimize_batch(gpu, batch)
    e = exp["patches"]
    ok_cpu = np.array([bool(p["ok"]) for p in e])
    assert np.array_equal(ok_cpu, batch.ok.astype(bool))
    both = np.nonzero(ok_cpu & batch.ok.astype(bool))[0]
    same = [e[k]["images"] == list(batch.images[k, :batch.n_images[k]]) and e[k]["nevals"] == batch.nevals[k] and
            np.array_equal(np.array(e[k]["center"], dtype=np.float32), batch.center[k]) and
            np.array_equal(np.array(e[k]["normal"], dtype=np.float32), batch.normal[k]) and
            np.array_equal(np.array(e[k]["color"], dtype=np.float32), batch.color[k]) for k in both]
    assert np.mean(same) >= 0.999, np.mean(same)
    gpu.close()


def test_init_patches_batch_matches_oracle(tiny_scene, oracle_scene, gpu_scene):
    """the Scene::initPatches seed loop (seed construction on the GPU, optimize, drift gate)
    against the oracle's restatement of Scene.cpp:112-178.  Bit-identical like optimize() itself."""
    from hpmvs_amd import api, synth
    from oracle import oracle as orc
    xyz, off, img = synth.make_nvm_points(tiny_scene, 400, start_level=2, noise=1.5)
    # make the early-exit branches occur: a point with too few measurements, one outside every image
    off = off.copy()
    batch = api.init_patches_batch(gpu_scene, xyz, off, img, start_level=2)
    P = orc.init_patches(oracle_scene, xyz, off, img, start_level=2, n_threads=8)
    st_cpu = np.array([p.stage for p in P])
    assert np.array_equal(st_cpu, batch.stage), np.nonzero(st_cpu != batch.stage)[0][:10]
    assert np.array_equal(st_cpu == 0, batch.ok.astype(bool))
    assert (st_cpu == 0).sum() >= 50 and (st_cpu == 12).sum() + (st_cpu >= 10).sum() >= 1
    for k in range(len(P)):
        p = P[k]
        assert np.array_equal(np.array(p.center[:], dtype=np.float32), batch.center[k]), k
        if p.stage < 10:  # seed was built: normal/scale/images defined
            assert np.float32(p.scale) == batch.scale[k]
            if p.stage == 0 or p.stage == 12:
                assert np.array_equal(np.array(p.normal[:], dtype=np.float32), batch.normal[k]), k
                assert list(p.images[:p.n_images]) == list(batch.images[k, :batch.n_images[k]])


def test_many_views_multi_chunk_is_bit_identical():
    """20 views, patches with 9-20 attached images: exercises the chunked evaluation (more than 8 images per
    patch), setINCCs with a non-zero reference slot and the O(V^2) setRefImage pass."""
    import torch
    from hpmvs_amd import api, synth
    from oracle import oracle as orc
    scene = synth.make_scene(20, 640, 360, n_waves=16, device=torch.device("cuda", 0))
    g = api.Scene(scene)
    osc = orc.OracleScene(scene)
    seeds = synth.make_seeds(scene, 160, start_level=1, max_images=64, max_seed_images=20, seed=synth.SEED + 5)
    idx = np.nonzero(seeds.n_images >= 9)[0]
    assert len(idx) >= 40
    batch = api.Batch.from_seeds(seeds, idx)
    P = orc.patches_from_seeds(seeds, idx)
    x0 = np.array([orc.initial_parameters(osc, P[k]) for k in range(len(idx))])
    fg, grabs = api.objective_batch(g, batch, x0)
    fc = np.array([orc.objective_at(osc, P[k], x0[k]) for k in range(len(idx))])
    assert np.array_equal(fg, fc) and grabs.max() > 8
    inc = api.inccs_batch(g, batch, 3, 1)
    for k in range(len(idx)):
        assert np.array_equal(inc[k, :P[k].n_images], orc.inccs(osc, P[k], 3, 1))
    b = api.Batch.from_seeds(seeds)
    api.optimize_batch(g, b)
    Q = orc.patches_from_seeds(seeds)
    orc.optimize_batch(osc, Q, n_threads=8)
    assert np.array_equal(np.array([q.stage for q in Q]), b.stage)
    both = np.nonzero(b.ok)[0]
    assert len(both) >= 100 and b.n_images[both].max() >= 9
    for k in both:
        assert np.array_equal(np.array(Q[k].center[:], dtype=np.float32), b.center[k])
        assert np.array_equal(np.array(Q[k].normal[:], dtype=np.float32), b.normal[k])
        assert list(Q[k].images[:Q[k].n_images]) == list(b.images[k, :b.n_images[k]]) and Q[k].nevals == b.nevals[k]
    g.close()


@pytest.mark.parametrize("opts", [
    dict(MIN_IMAGES_PER_PATCH=2, NCC_ALPHA_1=0.5, NCC_ALPHA_2=0.6),
    dict(MAX_ANGLE=float(np.float32(45.0 * np.pi / 180.0)), MIN_ANGLE=float(np.float32(5.0 * np.pi / 180.0)), NCC_ALPHA_1=0.3, NCC_ALPHA_2=0.4),
    # the level options: MAXLEVEL bounds sampleTexture's level (MAXLEVEL - 1) and addImages' (MAXLEVEL - 2); MINLEVEL addImages' floor
    dict(MAXLEVEL=4),
    dict(MAXLEVEL=3, NCC_ALPHA_1=0.3),
    dict(MINLEVEL=1, MAXLEVEL=5, MIN_IMAGES_PER_PATCH=2),
])
def test_non_default_options_are_honoured_identically(opts, tiny_seeds, oracle_scene, gpu_scene):
    """HpmvsOptions fields the path reads (reference include/hpmvs/HpmvsOptions.h:29-58) reach the kernel and
    the oracle alike: decisions, stages and refined patches stay identical, and differ from the defaults."""
    from hpmvs_amd import api
    from oracle import oracle as orc
    og, oc = api.default_options(), orc.default_options()
    for k, v in opts.items():
        setattr(og, k, v)
        setattr(oc, k, v)
    b = api.Batch.from_seeds(tiny_seeds)
    api.optimize_batch(gpu_scene, b, og)
    P = orc.patches_from_seeds(tiny_seeds)
    orc.optimize_batch(oracle_scene, P, options=oc, n_threads=8)
    stage_cpu = np.array([p.stage for p in P])
    assert np.array_equal(stage_cpu, b.stage)
    ok = np.nonzero(b.ok)[0]
    assert len(ok) >= (50 if ("MAXLEVEL" in opts or "MINLEVEL" in opts) else 20), len(ok)   # (the oracle refines 119-186 of the 200 under the level options)
    for k in ok:
        assert np.array_equal(np.array(P[k].center[:], dtype=np.float32), b.center[k])
        assert np.array_equal(np.array(P[k].normal[:], dtype=np.float32), b.normal[k])
        assert list(P[k].images[:P[k].n_images]) == list(b.images[k, :b.n_images[k]])
    d = api.Batch.from_seeds(tiny_seeds)
    api.optimize_batch(gpu_scene, d)
    assert not np.array_equal(d.stage, b.stage) or not np.array_equal(d.center, b.center)


def test_maximum_image_lists_and_overflow():
    """HPMVS_MAX_IMAGES = 64 attached images per patch: lists that reach the cap through addImages refine
    identically to the oracle; a patch handed over with more than the batch's max_images fails with stage 100
    and is left untouched (the reference's vector<int> is unbounded; the cap is the documented difference)."""
    import torch
    from hpmvs_amd import api, synth
    from oracle import oracle as orc
    scene = synth.make_scene(70, 480, 270, n_waves=12, device=torch.device("cuda", 0))
    g = api.Scene(scene)
    osc = orc.OracleScene(scene)
    seeds = synth.make_seeds(scene, 120, start_level=1, max_images=64, max_seed_images=60, seed=synth.SEED + 11)
    assert seeds.n_images.max() >= 30
    b = api.Batch.from_seeds(seeds)
    api.optimize_batch(g, b)
    Q = orc.patches_from_seeds(seeds)
    orc.optimize_batch(osc, Q, n_threads=8)
    assert np.array_equal(np.array([q.stage for q in Q]), b.stage)
    ok = np.nonzero(b.ok)[0]
    assert len(ok) >= 30
    for k in ok:
        assert np.array_equal(np.array(Q[k].center[:], dtype=np.float32), b.center[k])
        assert np.array_equal(np.array(Q[k].normal[:], dtype=np.float32), b.normal[k])
        assert list(Q[k].images[:Q[k].n_images]) == list(b.images[k, :b.n_images[k]]) and Q[k].nevals == b.nevals[k]
    # addImages can push a list past the cap: both sides then report stage 100
    print("max attached after refinement", int(b.n_images[ok].max()), "stage-100 patches", int((b.stage == 100).sum()))
    # a batch whose rows are narrower than a patch's list: stage 100, untouched
    narrow = api.Batch(seeds.center[:8], seeds.normal[:8], seeds.scale[:8], seeds.n_images[:8], seeds.images[:8, :16])
    narrow.n_images[:] = 20  # claims more images than the row holds
    api.optimize_batch(g, narrow)
    assert (narrow.ok == 0).all() and (narrow.stage == 100).all()
    assert np.array_equal(narrow.center, seeds.center[:8].astype(np.float32))
    g.close()


def test_invalid_image_ids_are_rejected_not_dereferenced(tiny_seeds, oracle_scene, gpu_scene):
    """An attached image id outside [0, n_views) fails that patch with stage 101 on both sides; the other
    patches of the batch are unaffected."""
    from hpmvs_amd import api
    from oracle import oracle as orc
    good = api.Batch.from_seeds(tiny_seeds)
    api.optimize_batch(gpu_scene, good)
    b = api.Batch.from_seeds(tiny_seeds)
    bad = [3, 17, 40]
    b.images[3, 0] = -1
    b.images[17, 1] = 1000
    b.images[40, 2] = gpu_scene.n_views if hasattr(gpu_scene, "n_views") else 3
    api.optimize_batch(gpu_scene, b)
    assert (b.ok[bad] == 0).all() and (b.stage[bad] == 101).all()
    rest = np.setdiff1d(np.arange(b.n), bad)
    assert np.array_equal(b.ok[rest], good.ok[rest]) and np.array_equal(b.center[rest], good.center[rest])
    P = orc.patches_from_seeds(tiny_seeds, bad)
    P[0].images[0] = -1; P[1].images[1] = 1000; P[2].images[2] = 3
    orc.optimize_batch(oracle_scene, P, n_threads=1)
    assert [p.stage for p in P] == [101, 101, 101]


def test_batch_size_does_not_matter(tiny_scene, gpu_scene):
    """The launch lays a batch out differently depending on its size (kernel_optimize.hip, `quota`): up to 112 patches per
    workgroup go to all eight wavefronts with fixed state slots (two wavefronts per region, no paging), a bigger share is
    packed into full wavefronts that time-share the regions, 4 or fewer patches of a host-pointer call go through the open
    batch, up to 4096 through the pooled staging.  A patch's result is the same bits whichever way its batch went."""
    from hpmvs_amd import api, synth
    n_big = 36000   # 141 patches per workgroup: packed wavefronts, LDS time-sharing
    seeds = synth.make_seeds(tiny_scene, n_big, start_level=2, max_images=16, seed=synth.SEED + 5)
    whole = api.Batch.from_seeds(seeds)
    api.optimize_batch(gpu_scene, whole)
    assert 0 < whole.ok.sum() < n_big
    fields = ("ok", "center", "normal", "n_images", "images", "color", "fmin", "x", "nevals", "stage", "ngrabs")
    for lo, n in ((0, 3), (100, 5), (1000, 700), (2000, 4096), (7000, 9000), (5000, 28000)):
        part = api.Batch.from_seeds(seeds, np.arange(lo, lo + n))
        api.optimize_batch(gpu_scene, part)
        for f in fields:
            assert np.array_equal(getattr(part, f), getattr(whole, f)[lo:lo + n]), (lo, n, f)


def test_batch_order_does_not_matter(tiny_scene, tiny_seeds, gpu_scene):
    """Patches are independent (PatchOptimizer.cpp:78-103): refining a permuted batch gives the permuted results,
    byte for byte -- the property the host-side tile sort (distributed.tile_order) and the patch-index
    sharding across GPUs rely on."""
    from hpmvs_amd import api
    from hpmvs_amd import distributed as dd
    P0 = np.array([list(api.camera_from_nvm(v.f, v.q, v.c, v.width, v.height, tiny_scene.max_level).P0)
                   for v in tiny_scene.views])
    perm = dd.tile_order(tiny_seeds.center, tiny_seeds.images, tiny_seeds.n_images, P0, tile_px=64)
    assert sorted(perm.tolist()) == list(range(len(perm))) and not np.array_equal(perm, np.arange(len(perm)))
    a = api.Batch.from_seeds(tiny_seeds)
    b = api.Batch.from_seeds(tiny_seeds, perm)
    api.optimize_batch(gpu_scene, a)
    api.optimize_batch(gpu_scene, b)
    for f in ("ok", "center", "normal", "n_images", "images", "color", "fmin", "x", "nevals", "stage", "ngrabs"):
        assert np.array_equal(getattr(a, f)[perm], getattr(b, f)), f
    # and the shards of a 3-way split are the corresponding slices of the whole
    n = a.n
    for r in range(3):
        lo, hi = dd.shard_range(n, r, 3)
        s = api.Batch.from_seeds(tiny_seeds, np.arange(lo, hi))
        api.optimize_batch(gpu_scene, s)
        for f in ("ok", "center", "normal", "n_images", "images", "fmin", "nevals", "stage"):
            assert np.array_equal(getattr(s, f), getattr(a, f)[lo:hi]), (r, f)


def test_both_builds_of_the_kernel_give_the_same_results(tiny_scene, tiny_seeds, gpu_scene, tmp_path):
    """The library holds two builds of the refinement kernel (29 patch slots per wavefront / 4 state regions per CU, all
    optimiser state in LDS; and 64 slots / 3 regions with the optimiser's big matrices and wsuma in global memory behind buffer
    loads, for batches from 768 patches per CU upwards, kernel_launch.hip).  The same seeds through each build,
    forced with HPMVS_SLOTS in separate processes: every output array is byte-identical, and equal to the in-process
    run (which the other tests compare with the oracle)."""
    import os, subprocess, sys
    from hpmvs_amd import api
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from hpmvs_amd import api, synth
scene = synth.make_scene(3, 640, 480, n_waves=24)
seeds = synth.make_seeds(scene, 200, start_level=2)
g = api.Scene(scene)
idx = np.tile(np.arange(len(seeds.scale)), 40)   # 8000 patches: every wavefront of a few workgroups is in use
b = api.Batch.from_seeds(seeds, idx)
api.optimize_batch(g, b)
np.savez(sys.argv[1], ok=b.ok, center=b.center, normal=b.normal, n_images=b.n_images, images=b.images, color=b.color,
         fmin=b.fmin, x=b.x, nevals=b.nevals, stage=b.stage, result=b.result)
''' % root
    outs = {}
    for slots in ("29", "64"):
        f = str(tmp_path / f"slots{slots}.npz")
        r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, HPMVS_SLOTS=slots), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[slots] = np.load(f)
    for k in outs["29"].files:
        assert np.array_equal(outs["29"][k], outs["64"][k]), k
    assert outs["29"]["ok"].sum() > 1000
    b = api.Batch.from_seeds(tiny_seeds, np.arange(len(tiny_seeds.scale)))
    api.optimize_batch(gpu_scene, b)
    n = len(tiny_seeds.scale)
    assert np.array_equal(outs["64"]["ok"][:n], b.ok) and np.array_equal(outs["64"]["center"][:n], b.center)
    assert np.array_equal(outs["64"]["nevals"][:n], b.nevals) and np.array_equal(outs["64"]["x"][:n], b.x)


def test_batch_launch_while_single_patch_callers_keep_an_open_batch_busy(tiny_seeds, gpu_scene):
    """A batch launch takes whole CUs (its workgroups use a CU's LDS), the open batch behind single-patch callers holds
    a quarter of every CU for as long as callers keep coming.  The batch launch tells the open batch to end instead of
    waiting for the callers to pause: it completes promptly while 16 threads call optimize-of-one in a loop, both
    give the results of an undisturbed run, and the callers carry on afterwards."""
    import threading, time
    from hpmvs_amd import api
    n = len(tiny_seeds.scale)
    ref = api.Batch.from_seeds(tiny_seeds, np.arange(n))
    api.optimize_batch(gpu_scene, ref)
    stop = threading.Event()
    errors, calls = [], [0] * 16

    def caller(k):
        i = 0
        try:
            while not stop.is_set():
                j = (k * 7 + i) % n
                b = api.Batch.from_seeds(tiny_seeds, np.array([j]))
                api.optimize_batch(gpu_scene, b)
                if b.ok[0] != ref.ok[j] or (b.ok[0] and not np.array_equal(b.center[0], ref.center[j])):
                    errors.append((k, j))
                i += 1
                calls[k] = i
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=caller, args=(k,)) for k in range(16)]
    for t in threads:
        t.start()
    time.sleep(0.3)
    before = sum(calls)
    big = api.Batch.from_seeds(tiny_seeds, np.tile(np.arange(n), 100))
    t0 = time.time()
    api.optimize_batch(gpu_scene, big)
    dt = time.time() - t0
    time.sleep(0.3)
    after = sum(calls)
    stop.set()
    for t in threads:
        t.join(timeout=60)
    assert not errors, errors[:3]
    assert dt < 20.0, dt
    assert before > 0 and after > before, (before, after)
    assert np.array_equal(big.ok.reshape(100, n), np.tile(ref.ok, (100, 1)))
    assert np.array_equal(big.center.reshape(100, n, 4)[37], ref.center)


def test_tree_sums_mode_is_opt_in_and_stays_inside_the_stated_tolerance(tiny_scene, tiny_seeds, oracle_scene, tmp_path):
    """HPMVS_SUMS=tree selects the throughput build of the kernel (every lane sums its window row first, the seven row
    sums are added in row order: photometric.hpp, HPMVS_FAST_SUMS).  It is NOT bit-identical to the oracle -- the
    sequential order of Patch2d.hpp:37-84 is what the default build keeps -- so it is held to the DISTRIBUTIONAL tolerance
    SURVEY 8d states for refinement against an arbitrary hpmvs binary: decisions >= 99 % equal on this
    200-seed scene, mean |d ncc| <= 5e-3, >= 80 % of the patches in the same basin.  The default (no variable, or
    `exact`) must be byte-identical to the in-process run that every other test compares with the oracle."""
    import os, subprocess, sys
    from hpmvs_amd import api
    from oracle import oracle as orc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from hpmvs_amd import api, synth
scene = synth.make_scene(3, 640, 480, n_waves=24)
seeds = synth.make_seeds(scene, 200, start_level=2)
g = api.Scene(scene)
b = api.Batch.from_seeds(seeds)
api.optimize_batch(g, b)
big = api.Batch.from_seeds(seeds, np.tile(np.arange(len(seeds.scale)), 40))   # 8000 patches: several wavefronts per workgroup
api.optimize_batch(g, big)
np.savez(sys.argv[1], ok=b.ok, center=b.center, normal=b.normal, fmin=b.fmin, nevals=b.nevals, stage=b.stage,
         big_ok=big.ok, big_center=big.center)
''' % root
    outs = {}
    for mode in ("exact", "tree"):
        f = str(tmp_path / f"sums_{mode}.npz")
        r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, HPMVS_SUMS=mode), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = np.load(f)
    n = len(tiny_seeds.scale)
    P = orc.patches_from_seeds(tiny_seeds)
    orc.optimize_batch(oracle_scene, P, which=orc.OPT_REF, n_threads=8)
    ok_cpu = np.array([p.stage == 0 for p in P])
    ex, tr = outs["exact"], outs["tree"]
    # exact: the oracle's decisions and bits
    assert np.array_equal(ex["ok"].astype(bool), ok_cpu)
    for k in np.nonzero(ok_cpu)[0]:
        assert np.array_equal(np.array(P[k].center[:], dtype=np.float32), ex["center"][k]), k
    # tree: inside the distributional tolerance
    ok_t = tr["ok"].astype(bool)
    agree = float((ok_t == ok_cpu).mean())
    both = np.nonzero(ok_t & ok_cpu)[0]
    unrob = lambda f: f / (1.0 - 3.0 * f)
    dncc = float(np.mean([abs(unrob(P[k].fmin) - unrob(tr["fmin"][k])) for k in both]))
    basin = []
    for k in both:
        dd = np.linalg.norm(np.array(P[k].center[:3]) - tr["center"][k, :3]) / float(tiny_seeds.scale[k])
        cs = float(np.clip(np.dot(np.array(P[k].normal[:3]), tr["normal"][k, :3]), -1, 1))
        basin.append(dd < 0.25 and np.degrees(np.arccos(cs)) < 2.0)
    assert agree >= 0.99 and dncc <= 5e-3 and np.mean(basin) >= 0.80, (agree, dncc, float(np.mean(basin)))
    assert not np.array_equal(tr["fmin"], ex["fmin"])  # it really is another build
    # the mode is deterministic across batch compositions, like the exact one
    assert np.array_equal(tr["big_ok"].reshape(40, n), np.tile(tr["ok"], (40, 1)))
    assert np.array_equal(tr["big_center"].reshape(40, n, 4)[23], tr["center"])
