"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Product code under hpmvs_amd/ must never import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MAX_IMAGES = 256  # = ORC_MAX_IMAGES = HPMVS_MAX_IMAGES
MAX_LEVELS = 8
OPT_REF, OPT_PORT = 0, 1


class Options(C.Structure):
    _fields_ = [("MAXLEVEL", C.c_int), ("MINLEVEL", C.c_int), ("MAX_ANGLE", C.c_float), ("MIN_ANGLE", C.c_float),
                ("MAX_IMAGES_PER_PATCH", C.c_int), ("MIN_IMAGES_PER_PATCH", C.c_int),
                ("NCC_ALPHA_1", C.c_float), ("NCC_ALPHA_2", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("n_levels", C.c_int), ("P", (C.c_float * 12) * MAX_LEVELS), ("K0", C.c_float * 9),
                ("center", C.c_float * 4), ("oaxis", C.c_float * 4), ("xaxis", C.c_float * 3),
                ("yaxis", C.c_float * 3), ("zaxis", C.c_float * 3)]


class Patch(C.Structure):
    _fields_ = [("center", C.c_float * 4), ("normal", C.c_float * 4), ("scale", C.c_float),
                ("n_images", C.c_int), ("images", C.c_int * MAX_IMAGES), ("ncc_const", C.c_float),
                ("color", C.c_float * 3), ("fmin", C.c_double), ("x", C.c_double * 3),
                ("nlopt_result", C.c_int), ("nevals", C.c_int), ("stage", C.c_int),
                ("ngrabs", C.c_long), ("ngrabs_all", C.c_long)]


OBJ_FN = C.CFUNCTYPE(C.c_double, C.c_uint, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)

_lib = None


def build(force: bool = False):
    """Compile oracle/liboracle.so (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(HERE, "liboracle.so")
    if force or not os.path.exists(so) or os.path.exists("/root/reference/thirdLibs/nlopt-2.4.2"):
        subprocess.run(["make", "-C", HERE, "-s"], check=True, capture_output=True)
    return so


def lib():
    global _lib
    if _lib is not None:
        return _lib
    so = os.path.join(HERE, "liboracle.so")
    if not os.path.exists(so):
        build()
    L = C.CDLL(so)
    L.orc_scene_create.restype = C.c_void_p
    L.orc_scene_create.argtypes = [C.c_int]
    L.orc_scene_destroy.argtypes = [C.c_void_p]
    L.orc_scene_set_view.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_double,
                                     C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
    L.orc_scene_set_covis.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]
    L.orc_scene_camera.restype = C.POINTER(Camera)
    L.orc_scene_camera.argtypes = [C.c_void_p, C.c_int]
    L.orc_scene_level.restype = C.POINTER(C.c_uint8)
    L.orc_scene_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_half_resize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.orc_default_options.argtypes = [C.POINTER(Options)]
    L.orc_optimizer_available.argtypes = [C.c_int]
    L.orc_optimize.argtypes = [C.c_void_p, C.POINTER(Options), C.POINTER(Patch), C.c_int]
    L.orc_optimize_batch.argtypes = [C.c_void_p, C.POINTER(Options), C.POINTER(Patch), C.c_int, C.c_int, C.c_int]
    L.orc_init_patches.argtypes = [C.c_void_p, C.POINTER(Options), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.POINTER(Patch), C.c_int, C.c_int]
    L.orc_objective_at.restype = C.c_double
    L.orc_objective_at.argtypes = [C.c_void_p, C.POINTER(Options), C.POINTER(Patch), C.POINTER(C.c_double)]
    L.orc_initial_parameters.argtypes = [C.c_void_p, C.POINTER(Options), C.POINTER(Patch), C.POINTER(C.c_double)]
    L.orc_inccs.argtypes = [C.c_void_p, C.POINTER(Options), C.POINTER(Patch), C.c_int, C.c_int,
                            C.POINTER(C.c_float)]
    L.orc_sample_texture.argtypes = [C.c_void_p, C.POINTER(Options), C.POINTER(Patch), C.c_int, C.c_int,
                                     C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_float)]
    L.orc_expand_batch.argtypes = [C.c_void_p, C.POINTER(Options), C.c_int, C.c_int, C.POINTER(Patch), C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.POINTER(Patch), C.c_int, C.c_int]
    L.orc_expand_directions.argtypes = [C.c_int, C.c_void_p]
    L.orc_depths_create.restype = C.c_void_p
    L.orc_depths_create.argtypes = [C.c_void_p]
    L.orc_depths_destroy.argtypes = [C.c_void_p]
    L.orc_depths_level.restype = C.POINTER(C.c_float)
    L.orc_depths_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_set_depths.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Patch)]
    L.orc_set_depths_ex.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Patch), C.c_int]
    L.orc_level_support.argtypes = [C.c_void_p, C.POINTER(Patch), C.c_int]
    L.orc_branch_round.argtypes = [C.c_void_p, C.POINTER(Options), C.c_void_p, C.c_int, C.POINTER(Patch), C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.POINTER(Patch), C.c_void_p, C.c_int]
    L.orc_cell_key.restype = C.c_longlong
    L.orc_cell_key.argtypes = [C.POINTER(C.c_float), C.c_float]
    L.orc_extend_round.argtypes = [C.c_void_p, C.POINTER(Options), C.c_void_p, C.c_int, C.POINTER(Patch), C.c_float, C.c_float,
                                   C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(Patch),
                                   C.POINTER(C.c_int), C.c_int]
    L.orc_depth_gates.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Patch), C.c_float, C.c_int, C.POINTER(C.c_int),
                                  C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_bobyqa_minimize.argtypes = [C.c_int, C.c_int, OBJ_FN, C.c_void_p, C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_int,
                                      C.POINTER(C.c_double), C.POINTER(C.c_int)]
    _lib = L
    return L


def default_options() -> Options:
    o = Options()
    lib().orc_default_options(C.byref(o))
    return o


def set_libm_mode(bits: int):
    """Diagnosis only (tools/full_parity.py): bit 0 = acos correctly rounded, bit 1 = the device's sin / cos series."""
    lib().orc_set_libm_mode(int(bits))


def set_dot4_order(order: int):
    """Evaluation order of Eigen's 4-float reductions in the oracle (0 left-to-right, 1 SSE2, 2 SSE3 haddps)."""
    lib().orc_set_dot4_order(int(order))


def set_vec_div_mode(mode: int):
    """Eigen's `vector / scalar` in the oracle: 0 true division (Eigen >= 3.2), 1 multiplication by the reciprocal
    (Eigen 3.0 / 3.1: normalized(), ave /= size3, result /= result(2))."""
    lib().orc_set_vec_div_mode(int(mode))


def optimizer_available(which: int) -> bool:
    return lib().orc_optimizer_available(which) == 0


def best_optimizer(allow_port: bool = False) -> int:
    """The optimiser every parity check runs: the reference's GENUINE NLopt BOBYQA (oracle/_ref).

    There is no silent downgrade: tests/native/bobyqa_port.cpp (libbq_port.so, loaded at run time, never linked into liboracle.so) is a host build of the PRODUCT's own
    bobyqa3.hpp, so comparing the HIP path against it proves nothing about the reference.  If
    oracle/_ref did not travel this raises; the port is only used where a caller asks for it by
    name (`which=OPT_PORT`, or allow_port=True / HPMVS_ORACLE_ALLOW_PORT=1 for timing-only use)."""
    if optimizer_available(OPT_REF):
        return OPT_REF
    if (allow_port or os.environ.get("HPMVS_ORACLE_ALLOW_PORT") == "1") and optimizer_available(OPT_PORT):
        import sys
        print("oracle: oracle/_ref/libref_bobyqa.so is missing -- using the BOBYQA PORT (a host build of the "
              "product's own state machine): results are NOT a parity check", file=sys.stderr)
        return OPT_PORT
    raise RuntimeError("oracle: oracle/_ref/libref_bobyqa.so (genuine NLopt BOBYQA, built by oracle/Makefile from "
                       "/root/reference in the build container) is missing; refusing to fall back to the port of the "
                       "product's own optimiser.  Compare against tests/golden instead.")


class OracleScene:
    """Scene::addCameras + extractCoVisiblilty state for the oracle (reference Scene.cpp:42-88, 241-298)."""

    def __init__(self, synth_scene):
        L = lib()
        self.n_views = synth_scene.n_views
        self.max_level = synth_scene.max_level
        self.h = L.orc_scene_create(self.n_views)
        for i, v in enumerate(synth_scene.views):
            rgb = v.rgb
            if not isinstance(rgb, np.ndarray):
                rgb = rgb.cpu().numpy()
            rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
            q = (C.c_double * 4)(*[float(t) for t in v.q])
            c = (C.c_double * 3)(*[float(t) for t in v.c])
            L.orc_scene_set_view(self.h, i, v.width, v.height, rgb.ctypes.data, float(v.f), q, c, self.max_level)
        for i, lst in enumerate(synth_scene.covis):
            arr = (C.c_int * max(1, len(lst)))(*lst)
            L.orc_scene_set_covis(self.h, i, arr, len(lst))

    def __del__(self):
        try:
            lib().orc_scene_destroy(self.h)
        except Exception:
            pass

    def camera(self, i) -> Camera:
        return lib().orc_scene_camera(self.h, i).contents

    def level(self, i, level) -> np.ndarray:
        w, h = C.c_int(), C.c_int()
        p = lib().orc_scene_level(self.h, i, level, C.byref(w), C.byref(h))
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()


class OracleDepths:
    """The depth maps of Scene (reference Scene.cpp:74-80) and the gates over them (Scene.cpp:351-381, 518-644)."""

    def __init__(self, scene: OracleScene):
        self.scene = scene
        self.h = lib().orc_depths_create(scene.h)

    def __del__(self):
        try:
            lib().orc_depths_destroy(self.h)
        except Exception:
            pass

    def level(self, view, level) -> np.ndarray:
        """One map as an array in Eigen's storage order: shape (cols, rows), element [x, y]."""
        r, c = C.c_int(), C.c_int()
        p = lib().orc_depths_level(self.h, view, level, C.byref(r), C.byref(c))
        return np.ctypeslib.as_array(p, shape=(c.value, r.value))

    def set_depths(self, patch, subtract=False):
        """Scene::setDepths(patch, subtract) (Scene.cpp:351-381)."""
        lib().orc_set_depths_ex(self.scene.h, self.h, C.byref(patch), 1 if subtract else 0)

    def gates(self, patch, margin=1.0, abs_int=0):
        v, b, f = C.c_int(), C.c_int(), C.c_int()
        lib().orc_depth_gates(self.scene.h, self.h, C.byref(patch), float(margin), int(abs_int), C.byref(v), C.byref(b), C.byref(f))
        return v.value, b.value, f.value


def patch_view(arr):
    """A numpy structured view (no copy) of a ctypes array of Patch: fields by the struct's own offsets."""
    import numpy as _np
    fmt = {"center": ("<f4", (4,)), "normal": ("<f4", (4,)), "scale": "<f4", "n_images": "<i4", "images": ("<i4", (MAX_IMAGES,)),
           "ncc_const": "<f4", "color": ("<f4", (3,)), "fmin": "<f8", "x": ("<f8", (3,)), "nlopt_result": "<i4", "nevals": "<i4",
           "stage": "<i4", "ngrabs": "<i8", "ngrabs_all": "<i8"}
    names = [n for n, _ in Patch._fields_]
    dt = _np.dtype(dict(names=names, formats=[fmt[n] for n in names], offsets=[getattr(Patch, n).offset for n in names],
                        itemsize=C.sizeof(Patch)))
    return _np.frombuffer(arr, dtype=dt)


def patches_from_seeds(seeds, idx=None):
    idx = range(len(seeds.scale)) if idx is None else idx
    arr = (Patch * len(idx))()
    for k, i in enumerate(idx):
        p = arr[k]
        p.center[:] = seeds.center[i].tolist()
        p.normal[:] = seeds.normal[i].tolist()
        p.scale = float(seeds.scale[i])
        n = int(seeds.n_images[i])
        p.n_images = n
        for j in range(n):
            p.images[j] = int(seeds.images[i, j])
    return arr


def optimize_batch(scene: OracleScene, patches, which=None, options=None, n_threads=1):
    L = lib()
    which = best_optimizer() if which is None else which
    o = options or default_options()
    L.orc_optimize_batch(scene.h, C.byref(o), patches, len(patches), which, n_threads)
    return patches


def init_patches(scene: OracleScene, xyz, meas_off, meas_img, start_level=4, which=None, options=None, n_threads=1):
    """Scene::initPatches seed loop (reference Scene.cpp:112-178)."""
    which = best_optimizer() if which is None else which
    o = options or default_options()
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    off = np.ascontiguousarray(meas_off, dtype=np.int32)
    img = np.ascontiguousarray(meas_img, dtype=np.int32)
    n = len(xyz)
    P = (Patch * n)()
    lib().orc_init_patches(scene.h, C.byref(o), start_level, n, xyz.ctypes.data, off.ctypes.data, img.ctypes.data, P,
                           which, n_threads)
    return P


def expand_directions(mode: int) -> np.ndarray:
    out = np.zeros(12, dtype=np.float32)
    n = lib().orc_expand_directions(mode, out.ctypes.data)
    return out[:2 * n].reshape(n, 2)


def expand_batch(scene: OracleScene, mode, parents, cell_center, cell_width, skip=None, which=None, options=None,
                 n_threads=1):
    """CellProcessor::extend / ::branch candidate loops (reference CellProcessor.cpp:84-178, 210-262)."""
    which = best_optimizer() if which is None else which
    o = options or default_options()
    n = len(parents)
    N = 6 if mode == 0 else 4
    cc = np.ascontiguousarray(cell_center, dtype=np.float32).reshape(n, 3)
    cw = np.ascontiguousarray(cell_width, dtype=np.float32).reshape(n)
    sk = None if skip is None else np.ascontiguousarray(skip, dtype=np.uint8).reshape(n * N)
    out = (Patch * (n * N))()
    lib().orc_expand_batch(scene.h, C.byref(o), mode, n, parents, cc.ctypes.data, cw.ctypes.data,
                           None if sk is None else sk.ctypes.data, out, which, n_threads)
    return out


def objective_at(scene: OracleScene, patch: Patch, x, options=None) -> float:
    o = options or default_options()
    xx = (C.c_double * 3)(*x)
    return lib().orc_objective_at(scene.h, C.byref(o), C.byref(patch), xx)


def initial_parameters(scene: OracleScene, patch: Patch, options=None):
    o = options or default_options()
    xx = (C.c_double * 3)()
    lib().orc_initial_parameters(scene.h, C.byref(o), C.byref(patch), xx)
    return [xx[0], xx[1], xx[2]]


def inccs(scene: OracleScene, patch: Patch, ref_idx=0, robust=0, options=None):
    o = options or default_options()
    out = (C.c_float * MAX_IMAGES)()
    lib().orc_inccs(scene.h, C.byref(o), C.byref(patch), ref_idx, robust, out)
    return np.array(out[:patch.n_images], dtype=np.float32)


def sample_texture(scene: OracleScene, patch: Patch, ref_slot, slot, options=None):
    o = options or default_options()
    tex = (C.c_float * 147)()
    lvl = C.c_int()
    geo = (C.c_float * 6)()
    ok = lib().orc_sample_texture(scene.h, C.byref(o), C.byref(patch), ref_slot, slot, tex, C.byref(lvl), geo)
    return bool(ok), np.array(tex, dtype=np.float32), lvl.value, np.array(geo, dtype=np.float32)


def half_resize(img: np.ndarray) -> np.ndarray:
    h, w, _ = img.shape
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.zeros((h // 2, w // 2, 3), dtype=np.uint8)
    lib().orc_half_resize(img.ctypes.data, w, h, out.ctypes.data)
    return out


# ---- the reference's GENUINE CImg pyramid (oracle/_ref/libref_cimg.so, built by oracle/Makefile from
# /root/reference/thirdLibs/cimg/CImg.h; travels to the GPU box as a prebuilt checker like libref_bobyqa.so)
_cimg = None


def cimg_available() -> bool:
    return os.path.exists(os.path.join(HERE, "_ref", "libref_cimg.so"))


def cimg_pyramid(img: np.ndarray, n_levels: int):
    """Image::load's pyramid (reference src/hpmvs/Image.cpp:55-63) by the genuine CImg: list of
    interleaved uint8 [h, w, 3] arrays, level 0 first."""
    global _cimg
    if _cimg is None:
        _cimg = C.CDLL(os.path.join(HERE, "_ref", "libref_cimg.so"))
        _cimg.ref_cimg_pyramid.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                           C.POINTER(C.c_int), C.POINTER(C.c_int)]
    h, w, _ = img.shape
    img = np.ascontiguousarray(img, dtype=np.uint8)
    outs, ws, hs = [], (C.c_int * n_levels)(), (C.c_int * n_levels)()
    ww, hh = w, h
    for _ in range(n_levels):
        outs.append(np.zeros((hh, ww, 3), dtype=np.uint8))
        ww, hh = ww // 2, hh // 2
    ptrs = (C.c_void_p * n_levels)(*[o.ctypes.data for o in outs])
    rc = _cimg.ref_cimg_pyramid(img.ctypes.data, w, h, n_levels, ptrs, ws, hs)
    if rc != 0:
        raise RuntimeError("ref_cimg_pyramid failed")
    for l in range(n_levels):
        assert (ws[l], hs[l]) == (outs[l].shape[1], outs[l].shape[0]), "CImg level size differs from floor(w/2) x floor(h/2)"
    return outs


def cell_key(p, width) -> int:
    """Occupancy key of the grid cell (floor(p / width) per axis) that holds the point p."""
    a = (C.c_float * 3)(float(p[0]), float(p[1]), float(p[2]))
    return int(lib().orc_cell_key(a, float(width)))


def extend_round(scene: OracleScene, depths: OracleDepths, parents, width, occupied, margin=1.0, abs_int=0, frozen_gates=False,
                 which=None, options=None):
    """One priority level of the expansion, sequentially (CellProcessor::extend over the leaves in order, reference
    CellProcessor.cpp:84-178), on a caller-owned occupancy set (a Python set of cell keys, updated in place) and the
    depth maps `depths` (updated in place).  Returns (candidates[6 n], counts[6 n, 3])."""
    which = best_optimizer() if which is None else which
    o = options or default_options()
    n = len(parents)
    cap = len(occupied) + 6 * n + 1
    keys = (C.c_longlong * cap)(*sorted(occupied))
    n_occ = C.c_int(len(occupied))
    out = (Patch * (6 * n))()
    counts = (C.c_int * (18 * n))()
    lib().orc_extend_round(scene.h, C.byref(o), depths.h, n, parents, float(width), float(margin), int(abs_int), keys,
                           C.byref(n_occ), cap, 1 if frozen_gates else 0, out, counts, which)
    occupied.clear()
    occupied.update(int(keys[i]) for i in range(n_occ.value))
    return out, np.array(counts[:], dtype=np.int32).reshape(6 * n, 3)


def level_support(scene: OracleScene, patch, min_level=0) -> int:
    """Scene::getLevelSupport (Scene.cpp:334-343)."""
    return int(lib().orc_level_support(scene.h, C.byref(patch), int(min_level)))


def branch_round(scene: OracleScene, depths: OracleDepths, parents, cell_center, cell_width, final_level, which=None, options=None):
    """One priority level of CellProcessor::branch, leaf by leaf (reference CellProcessor.cpp:210-307) on the depth maps
    `depths` (updated in place).  Returns (candidates[4 n], split[n])."""
    which = best_optimizer() if which is None else which
    o = options or default_options()
    n = len(parents)
    cc = np.ascontiguousarray(cell_center, dtype=np.float32).reshape(n, 3)
    cw = np.ascontiguousarray(cell_width, dtype=np.float32).reshape(n)
    fl = np.ascontiguousarray(final_level, dtype=np.uint8).reshape(n)
    out = (Patch * (4 * n))()
    split = np.zeros(n, np.uint8)
    lib().orc_branch_round(scene.h, C.byref(o), depths.h, n, parents, cc.ctypes.data, cw.ctypes.data, fl.ctypes.data, out,
                           split.ctypes.data, which)
    return out, split
