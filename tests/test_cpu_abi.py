"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/hpmvs_amd.h declares, its host-side helpers work without a GPU, and every compute entry
point refuses to run without one (there is no CPU fallback to fall into)."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "hpmvs_amd.h")).read()
    return sorted(set(re.findall(r"\b(hpmvs_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from hpmvs_amd import api
    L = api.lib()
    syms = declared_symbols()
    assert len(syms) >= 16
    for s in syms:
        assert hasattr(L, s), s
    assert set(api.EXPORTS) == set(syms)


def test_camera_from_nvm_matches_oracle_bit_for_bit(tiny_scene, oracle_scene):
    """Two independent restatements of Camera::init (Camera.cpp:34-81)."""
    from hpmvs_amd import api
    for i, v in enumerate(tiny_scene.views):
        cam = api.camera_from_nvm(v.f, v.q, v.c, v.width, v.height, 5)
        ref = oracle_scene.camera(i)
        assert list(cam.P0) == list(ref.P[0])
        assert list(cam.center) == list(ref.center)
        assert list(cam.xaxis) == list(ref.xaxis) and list(cam.yaxis) == list(ref.yaxis)
        assert list(cam.zaxis) == list(ref.zaxis)
        assert cam.fsum == ref.K0[0] + ref.K0[4] and cam.n_levels == 6


def test_default_options_match_reference_header():
    from hpmvs_amd import api
    o = api.default_options()
    assert (o.MAXLEVEL, o.MINLEVEL, o.MIN_IMAGES_PER_PATCH, o.MAX_IMAGES_PER_PATCH) == (5, 0, 3, 6)
    assert abs(o.MAX_ANGLE - np.float32(np.pi / 3)) < 1e-6 and abs(o.MIN_ANGLE - np.float32(np.pi / 18)) < 1e-6
    assert o.NCC_ALPHA_1 == np.float32(0.4) and o.NCC_ALPHA_2 == np.float32(0.5)


def test_no_cpu_fallback_without_a_device():
    from hpmvs_amd import api
    L = api.lib()
    if api.device_count() > 0:
        return  # on a GPU box the gpu-marked tests cover the compute calls
    h = C.c_void_p()
    assert L.hpmvs_scene_create(3, 0, C.byref(h)) == -4  # HPMVS_ERR_NODEVICE
    assert b"no HIP device" in L.hpmvs_last_error()
    img = np.zeros((8, 8, 3), np.uint8)
    out = np.zeros((4, 4, 3), np.uint8)
    assert L.hpmvs_build_pyramid(0, img.ctypes.data, 8, 8, out.ctypes.data, 0) == -4
    o = api.default_options()
    b = api.PatchBatch()
    assert L.hpmvs_optimize_batch(None, C.byref(o), C.byref(b), 0, None) == -2  # null scene: HPMVS_ERR_ARG


def test_expand_direction_table_matches_libm():
    from hpmvs_amd import api
    """The candidate directions of CellProcessor::extend / ::branch are std::cos/std::sin of a float angle
    (reference CellProcessor.cpp:107-109, 233-235).  The kernels use a literal table; it must equal what the
    host C library returns (the oracle calls cosf/sinf)."""
    from oracle import oracle as orc
    for mode, n in ((api.EXPAND_EXTEND, 6), (api.EXPAND_BRANCH, 4)):
        t = api.expand_directions(mode)
        assert t.shape == (n, 2)
        assert np.array_equal(t, orc.expand_directions(mode))


def test_allgather_records_failure_paths_without_rccl():
    """hpmvs_allgather_records before any collective can start: a null communicator is an argument error;
    when no ncclAllGather can be found (HPMVS_RCCL_LIBRARY pointing at nothing) the call reports HPMVS_ERR_STATE with
    a message naming the look-up, it does not crash.  (Separate process: the look-up is cached.)"""
    import subprocess, sys, os
    code = r'''
import ctypes as C, sys
sys.path.insert(0, %r)
from hpmvs_amd import api
L = api.lib()
L.hpmvs_allgather_records.restype = C.c_int
L.hpmvs_allgather_records.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
buf = (C.c_uint8 * 192)()
r0 = L.hpmvs_allgather_records(None, buf, 1, buf, None)
m0 = L.hpmvs_last_error()
r1 = L.hpmvs_allgather_records(C.c_void_p(0x1000), buf, 1, buf, None)
m1 = L.hpmvs_last_error()
print(r0, m0.decode()); print(r1, m1.decode())
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HPMVS_RCCL_LIBRARY="/nonexistent/librccl_missing.so")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    l0, l1 = out.stdout.strip().splitlines()[-2:]
    assert l0.startswith("-2 ") and "null argument" in l0      # HPMVS_ERR_ARG
    assert l1.startswith("-3 ") and "HPMVS_RCCL_LIBRARY" in l1  # HPMVS_ERR_STATE
