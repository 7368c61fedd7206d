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
    bad = b.c_struct(); bad.max_images = 65
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
