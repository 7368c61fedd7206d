"""The C ABI reports misuse as negative status codes with a message (never a crash, never silent success):
scene life cycle, argument validation, option ranges (include/hpmvs_amd.h)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_scene_life_cycle_and_argument_errors(tiny_scene, tiny_seeds):
    from hpmvs_amd import api
    L = api.lib()
    h = C.c_void_p()
    assert L.hpmvs_scene_create(0, 0, C.byref(h)) == -2            # HPMVS_ERR_ARG: no views
    assert L.hpmvs_scene_create(2, 99, C.byref(h)) == -2           # bad device index
    assert L.hpmvs_scene_create(2, 0, C.byref(h)) == 0
    assert L.hpmvs_scene_commit(h) == -3                           # HPMVS_ERR_STATE: views not set
    assert b"not set" in L.hpmvs_last_error()
    v = tiny_scene.views[0]
    rgb = np.ascontiguousarray(v.rgb if isinstance(v.rgb, np.ndarray) else v.rgb.cpu().numpy(), dtype=np.uint8)
    cam = api.camera_from_nvm(v.f, v.q, v.c, v.width, v.height)
    assert L.hpmvs_scene_set_view(h, 5, v.width, v.height, rgb.ctypes.data, 0, C.byref(cam)) == -2   # view index
    assert L.hpmvs_scene_set_view(h, 0, v.width, v.height, rgb.ctypes.data, 0, C.byref(cam)) == 0
    assert L.hpmvs_scene_set_view(h, 1, v.width, v.height, rgb.ctypes.data, 0, C.byref(cam)) == 0
    ids = (C.c_int32 * 2)(0, 7)
    assert L.hpmvs_scene_set_covis(h, 0, ids, 2) == -2             # covisible id outside the scene
    ids = (C.c_int32 * 1)(1)
    assert L.hpmvs_scene_set_covis(h, 0, ids, 1) == 0
    # compute before commit
    b = api.Batch.from_seeds(tiny_seeds, np.arange(4))
    o = api.default_options()
    cb = b.c_struct()
    assert L.hpmvs_optimize_batch(h, C.byref(o), C.byref(cb), 0, None) == -3
    assert L.hpmvs_scene_commit(h) == 0
    assert L.hpmvs_scene_commit(h) == -3                           # twice
    assert L.hpmvs_scene_set_covis(h, 0, ids, 1) == -3             # after commit
    # bad batches / options
    bad = b.c_struct(); bad.max_images = 257   # > HPMVS_MAX_IMAGES
    assert L.hpmvs_optimize_batch(h, C.byref(o), C.byref(bad), 0, None) == -2
    bad = b.c_struct(); bad.ok = None
    assert L.hpmvs_optimize_batch(h, C.byref(o), C.byref(bad), 0, None) == -2
    o2 = api.default_options(); o2.MAXLEVEL = 9
    assert L.hpmvs_optimize_batch(h, C.byref(o2), C.byref(cb), 0, None) == -2
    assert L.hpmvs_optimize_batch(None, C.byref(o), C.byref(cb), 0, None) == -2
    # a correct call still works on this 2-view scene (all patches fail for lack of images, none crashes)
    assert L.hpmvs_optimize_batch(h, C.byref(o), C.byref(cb), 0, None) == 0
    assert (b.ok == 0).all()
    assert L.hpmvs_scene_destroy(h) == 0


def test_options_deeper_than_the_pyramid_are_refused(tiny_scene, tiny_seeds):
    """sampleTexture reads levels up to MAXLEVEL - 1 and the device forms the level's address arithmetically;
    a scene whose views hold fewer pyramid levels than options.MAXLEVEL must be refused, not sampled past its slab."""
    import copy
    from hpmvs_amd import api
    L = api.lib()
    shallow = copy.copy(tiny_scene)
    shallow.max_level = 2              # 3 levels per view
    sc = api.Scene(shallow, device=0)
    b = api.Batch.from_seeds(tiny_seeds, np.arange(8))
    cb = b.c_struct()
    o = api.default_options()          # MAXLEVEL = 5 > 3 levels
    assert L.hpmvs_optimize_batch(sc.h, C.byref(o), C.byref(cb), 0, None) == -2
    assert b"MAXLEVEL" in L.hpmvs_last_error()
    o.MAXLEVEL = 3                     # exactly as deep as the pyramid: served
    assert L.hpmvs_optimize_batch(sc.h, C.byref(o), C.byref(cb), 0, None) == 0
    sc.close()


def test_expand_shape_errors(tiny_seeds, gpu_scene):
    from hpmvs_amd import api
    L = api.lib()
    par = api.Batch.from_seeds(tiny_seeds, np.arange(4))
    out = api.Batch(np.zeros((20, 4), np.float32), np.zeros((20, 4), np.float32), np.zeros(20, np.float32),
                    np.zeros(20, np.int32), np.full((20, par.max_images), -1, np.int32))
    pb, ob, o = par.c_struct(), out.c_struct(), api.default_options()
    cc = np.zeros((4, 3), np.float32); cw = np.ones(4, np.float32)
    assert L.hpmvs_expand_batch(gpu_scene.h, C.byref(o), 0, C.byref(pb), cc.ctypes.data, cw.ctypes.data, None, C.byref(ob), 0, None) == -2
    assert b"fan-out" in L.hpmvs_last_error()


def test_bad_image_lists_reach_the_standalone_kernels_as_values_not_faults(tiny_seeds, gpu_scene):
    """hpmvs_objective_batch / hpmvs_inccs_batch take image ids straight from the caller.  An id outside
    the scene, a -1 padded entry inside the counted part, or a count beyond the row is answered with the plateau
    value 2.0 (objective_fn / setINCCs when the reference grab fails) -- no out-of-bounds read, no GPU fault."""
    from hpmvs_amd import api
    idx = np.nonzero(tiny_seeds.n_images >= 3)[0][:6]
    good = api.Batch.from_seeds(tiny_seeds, idx)
    x = np.zeros((good.n, 3))
    f_good, _ = api.objective_batch(gpu_scene, good, x)
    bad = api.Batch.from_seeds(tiny_seeds, idx)
    bad.images[0, 1] = 9999          # outside the scene
    bad.images[1, 0] = -1            # padding marker in the reference slot
    bad.n_images[2] = bad.max_images + 5   # longer than the row
    bad.n_images[3] = 0
    f_bad, g_bad = api.objective_batch(gpu_scene, bad, x)
    assert np.all(f_bad[:4] == 2.0) and np.all(g_bad[:4] == 0)
    assert np.array_equal(f_bad[4:], f_good[4:])      # the untouched patches are unaffected
    inc = api.inccs_batch(gpu_scene, bad, 0, 0)
    for k in range(3):
        nim = min(int(bad.n_images[k]), bad.max_images)
        assert np.all(inc[k, :nim] == 2.0)
    # and the gpu is still alive
    f_again, _ = api.objective_batch(gpu_scene, good, x)
    assert np.array_equal(f_again, f_good)


def test_option_ranges_are_validated(tiny_seeds, gpu_scene):
    from hpmvs_amd import api
    L = api.lib()
    b = api.Batch.from_seeds(tiny_seeds, np.arange(4))
    cb = b.c_struct()
    for field, value in (("MINLEVEL", -1), ("MINLEVEL", 6), ("MIN_IMAGES_PER_PATCH", 0), ("MAXLEVEL", 0)):
        o = api.default_options()
        setattr(o, field, value)
        assert L.hpmvs_optimize_batch(gpu_scene.h, C.byref(o), C.byref(cb), 0, None) == -2, field
        assert field.encode() in L.hpmvs_last_error()


def test_init_patches_without_any_measurement(tiny_scene, gpu_scene):
    """meas_off all zero and meas_img NULL: every seed fails with stage 10 (empty image list, Scene.cpp:152) instead
    of the call being rejected."""
    from hpmvs_amd import api
    xyz = np.zeros((5, 3))
    batch = api.init_patches_batch(gpu_scene, xyz, np.zeros(6, np.int32), np.zeros(0, np.int32), start_level=2)
    assert (batch.ok == 0).all() and (batch.stage == 10).all()
