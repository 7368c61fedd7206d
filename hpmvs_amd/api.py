"""ctypes binding of the product C ABI (include/hpmvs_amd.h -> hpmvs_amd/libhpmvs_amd.so).

Plumbing only: it loads the in-tree shared library (built by `__graft_entry__.build()` /
`make -C hpmvs_amd/csrc`), marshals numpy arrays or raw device pointers, and raises on any
non-zero status.  There is no Python or CPU implementation of the hot path behind it: if the
library or a GPU is missing the calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

try:  # torch first: it must be the one to load libamdhip64 ("one HIP runtime")
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional plumbing
    torch = None

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HPMVS_LIB") or os.path.join(HERE, "libhpmvs_amd.so")   # (HPMVS_LIB: a variant build, measurements only)
MAX_IMAGES = 256   # HPMVS_MAX_IMAGES: ids per list row the C ABI takes
FAST_IMAGES = 64   # HPMVS_FAST_IMAGES: lists that never outgrow this run in the batch kernel alone
MAX_LEVELS = 8


class HpmvsError(RuntimeError):
    pass


class Options(C.Structure):
    _fields_ = [("MAXLEVEL", C.c_int32), ("MINLEVEL", C.c_int32), ("MAX_ANGLE", C.c_float),
                ("MIN_ANGLE", C.c_float), ("MAX_IMAGES_PER_PATCH", C.c_int32),
                ("MIN_IMAGES_PER_PATCH", C.c_int32), ("NCC_ALPHA_1", C.c_float), ("NCC_ALPHA_2", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("P0", C.c_float * 12), ("center", C.c_float * 4), ("xaxis", C.c_float * 3),
                ("yaxis", C.c_float * 3), ("zaxis", C.c_float * 3), ("fsum", C.c_float), ("n_levels", C.c_int32)]


class PatchBatch(C.Structure):
    _fields_ = [("n", C.c_int32), ("max_images", C.c_int32), ("center", C.c_void_p), ("normal", C.c_void_p),
                ("scale", C.c_void_p), ("n_images", C.c_void_p), ("images", C.c_void_p), ("ok", C.c_void_p),
                ("color", C.c_void_p), ("ncc", C.c_void_p), ("fmin", C.c_void_p), ("x", C.c_void_p),
                ("result", C.c_void_p), ("nevals", C.c_void_p), ("stage", C.c_void_p), ("ngrabs", C.c_void_p)]


EXPORTS = [
    "hpmvs_last_error", "hpmvs_device_count", "hpmvs_build_id", "hpmvs_default_options", "hpmvs_camera_from_nvm",
    "hpmvs_scene_create", "hpmvs_scene_set_view", "hpmvs_scene_set_covis", "hpmvs_scene_commit",
    "hpmvs_scene_destroy", "hpmvs_scene_get_level", "hpmvs_scene_bytes", "hpmvs_build_pyramid",
    "hpmvs_optimize_batch", "hpmvs_init_patches_batch", "hpmvs_objective_batch", "hpmvs_inccs_batch", "hpmvs_last_kernel_ms",
    "hpmvs_last_wide_patches",
    "hpmvs_expand_batch", "hpmvs_expand_directions", "hpmvs_selftest_bobyqa",
    "hpmvs_pack_records", "hpmvs_unpack_records", "hpmvs_allgather_records", "hpmvs_pack_record_tails", "hpmvs_unpack_record_tails",
    "hpmvs_scene_depth_reset", "hpmvs_scene_depth_set_level", "hpmvs_scene_depth_get_level",
    "hpmvs_set_depths_batch", "hpmvs_depth_gates_batch", "hpmvs_depth_footprints_batch", "hpmvs_depth_ops_batch", "hpmvs_level_support_batch",
    "hpmvs_host_alloc", "hpmvs_host_free", "hpmvs_last_staging",
]

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HpmvsError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(there is no fallback implementation)")
    L = C.CDLL(LIB_PATH)
    L.hpmvs_last_error.restype = C.c_char_p
    L.hpmvs_build_id.restype = C.c_char_p
    L.hpmvs_default_options.argtypes = [C.POINTER(Options)]
    L.hpmvs_camera_from_nvm.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_int,
                                        C.c_int, C.POINTER(Camera)]
    L.hpmvs_scene_create.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.hpmvs_scene_set_view.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(Camera)]
    L.hpmvs_scene_set_covis.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.c_int]
    L.hpmvs_scene_commit.argtypes = [C.c_void_p]
    L.hpmvs_scene_destroy.argtypes = [C.c_void_p]
    L.hpmvs_scene_get_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int),
                                        C.POINTER(C.c_int)]
    L.hpmvs_scene_bytes.restype = C.c_size_t
    L.hpmvs_scene_bytes.argtypes = [C.c_void_p]
    L.hpmvs_build_pyramid.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    L.hpmvs_optimize_batch.argtypes = [C.c_void_p, C.POINTER(Options), C.POINTER(PatchBatch), C.c_int, C.c_void_p]
    L.hpmvs_init_patches_batch.argtypes = [C.c_void_p, C.POINTER(Options), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.POINTER(PatchBatch), C.c_int, C.c_void_p]
    L.hpmvs_objective_batch.argtypes = [C.c_void_p, C.POINTER(Options), C.POINTER(PatchBatch), C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.hpmvs_inccs_batch.argtypes = [C.c_void_p, C.POINTER(Options), C.POINTER(PatchBatch), C.c_int, C.c_int,
                                    C.c_void_p, C.c_int, C.c_void_p]
    L.hpmvs_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.hpmvs_host_alloc.argtypes = [C.c_size_t]
    L.hpmvs_host_alloc.restype = C.c_void_p
    L.hpmvs_host_free.argtypes = [C.c_void_p]
    L.hpmvs_host_free.restype = None
    L.hpmvs_last_staging.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    L.hpmvs_expand_batch.argtypes = [C.c_void_p, C.POINTER(Options), C.c_int, C.POINTER(PatchBatch), C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.POINTER(PatchBatch), C.c_int, C.c_void_p]
    L.hpmvs_expand_directions.argtypes = [C.c_int, C.c_void_p]
    L.hpmvs_scene_depth_reset.argtypes = [C.c_void_p]
    L.hpmvs_scene_depth_set_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.hpmvs_scene_depth_get_level.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_int),
                                              C.POINTER(C.c_int)]
    L.hpmvs_set_depths_batch.argtypes = [C.c_void_p, C.POINTER(PatchBatch), C.c_int, C.c_void_p]
    L.hpmvs_depth_footprints_batch.argtypes = [C.c_void_p, C.POINTER(PatchBatch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_int, C.c_void_p]
    L.hpmvs_depth_gates_batch.argtypes = [C.c_void_p, C.POINTER(PatchBatch), C.c_float, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int, C.c_void_p]
    L.hpmvs_depth_ops_batch.argtypes = [C.c_void_p, C.POINTER(PatchBatch), C.c_void_p, C.c_int, C.c_void_p]
    L.hpmvs_level_support_batch.argtypes = [C.c_void_p, C.POINTER(PatchBatch), C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    _lib = L
    return L


def _chk(rc):
    if rc != 0:
        raise HpmvsError(f"hpmvs_amd error {rc}: {lib().hpmvs_last_error().decode()}")


def build_id() -> str:
    """Source hash the loaded library was built from (stamped by hpmvs_amd/csrc/Makefile)."""
    return lib().hpmvs_build_id().decode()


def device_count() -> int:
    return lib().hpmvs_device_count()


def default_options() -> Options:
    o = Options()
    lib().hpmvs_default_options(C.byref(o))
    return o


def camera_from_nvm(f, q, c, width, height, max_level=5) -> Camera:
    cam = Camera()
    qq = (C.c_double * 4)(*[float(t) for t in q])
    cc = (C.c_double * 3)(*[float(t) for t in c])
    _chk(lib().hpmvs_camera_from_nvm(float(f), qq, cc, int(width), int(height), int(max_level), C.byref(cam)))
    return cam


def _ptr(a):
    """numpy array / torch tensor / int -> raw address (or None)."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    if isinstance(a, int):
        return a
    return a.data_ptr()  # torch tensor


class Scene:
    """HBM-resident scene: Scene::addCameras + extractCoVisiblilty state the path reads
    (reference include/hpmvs/Scene.h:69-71)."""

    def __init__(self, synth_scene, device: int = 0):
        L = lib()
        self.h = C.c_void_p()
        self.device = device
        self.n_views = synth_scene.n_views
        _chk(L.hpmvs_scene_create(self.n_views, device, C.byref(self.h)))
        for i, v in enumerate(synth_scene.views):
            cam = camera_from_nvm(v.f, v.q, v.c, v.width, v.height, synth_scene.max_level)
            rgb = v.rgb
            if isinstance(rgb, np.ndarray):
                rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
                _chk(L.hpmvs_scene_set_view(self.h, i, v.width, v.height, rgb.ctypes.data, 0, C.byref(cam)))
            else:  # torch tensor already in HBM
                rgb = rgb.contiguous()
                _chk(L.hpmvs_scene_set_view(self.h, i, v.width, v.height, rgb.data_ptr(), 1, C.byref(cam)))
        for i, lst in enumerate(synth_scene.covis):
            arr = (C.c_int32 * max(1, len(lst)))(*lst)
            _chk(L.hpmvs_scene_set_covis(self.h, i, arr, len(lst)))
        _chk(L.hpmvs_scene_commit(self.h))

    def close(self):
        if self.h:
            lib().hpmvs_scene_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bytes(self) -> int:
        return lib().hpmvs_scene_bytes(self.h)

    def level(self, view: int, level: int) -> np.ndarray:
        w, h = C.c_int(), C.c_int()
        _chk(lib().hpmvs_scene_get_level(self.h, view, level, None, 0, C.byref(w), C.byref(h)))
        out = np.empty((h.value, w.value, 3), dtype=np.uint8)
        _chk(lib().hpmvs_scene_get_level(self.h, view, level, out.ctypes.data, out.nbytes, C.byref(w), C.byref(h)))
        return out

    def last_staging(self):
        """(bytes copied through device buffers, bytes used in place) of the last host-pointer optimize_batch call."""
        a, b = C.c_ulonglong(), C.c_ulonglong()
        _chk(lib().hpmvs_last_staging(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def last_wide_patches(self) -> int:
        """Patches of the last refinement launch whose list outgrew 64 ids and that the wide kernel redid."""
        n = C.c_int32()
        _chk(lib().hpmvs_last_wide_patches(self.h, C.byref(n)))
        return n.value

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        _chk(lib().hpmvs_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value


class Batch:
    """Host-side (numpy) structure-of-arrays Patch3d batch."""

    def __init__(self, center, normal, scale, n_images, images):
        self.n = int(len(scale))
        self.max_images = int(images.shape[1])
        self.center = np.ascontiguousarray(center, dtype=np.float32).copy()
        self.normal = np.ascontiguousarray(normal, dtype=np.float32).copy()
        self.scale = np.ascontiguousarray(scale, dtype=np.float32).copy()
        self.n_images = np.ascontiguousarray(n_images, dtype=np.int32).copy()
        self.images = np.ascontiguousarray(images, dtype=np.int32).copy()
        n = self.n
        self.ok = np.zeros(n, dtype=np.uint8)
        self.color = np.zeros((n, 3), dtype=np.float32)
        self.ncc = np.zeros(n, dtype=np.float32)
        self.fmin = np.zeros(n, dtype=np.float64)
        self.x = np.zeros((n, 3), dtype=np.float64)
        self.result = np.zeros(n, dtype=np.int32)
        self.nevals = np.zeros(n, dtype=np.int32)
        self.stage = np.zeros(n, dtype=np.int32)
        self.ngrabs = np.zeros(n, dtype=np.int32)

    @classmethod
    def from_seeds(cls, seeds, idx=None):
        if idx is None:
            idx = np.arange(len(seeds.scale))
        idx = np.asarray(idx)
        return cls(seeds.center[idx], seeds.normal[idx], seeds.scale[idx], seeds.n_images[idx], seeds.images[idx])

    FIELDS = ("center", "normal", "scale", "n_images", "images", "ok", "color", "ncc", "fmin", "x", "result", "nevals",
              "stage", "ngrabs")

    def pin(self):
        """Move every array into pinned host memory (hpmvs_host_alloc).  Host-pointer calls then use the arrays in place
        (include/hpmvs_amd.h, "Host-pointer calls and pinned memory").  A pinned block lives exactly as long as the last
        numpy view of it: `c = batch.center` stays valid after the Batch is gone, and pinning again releases the previous
        blocks as soon as nothing refers to them."""
        import weakref
        L = lib()
        for name in self.FIELDS:
            a = getattr(self, name)
            nbytes = max(int(a.nbytes), 1)
            p = L.hpmvs_host_alloc(nbytes)
            if not p:
                raise MemoryError(f"hpmvs_host_alloc({nbytes})")
            buf = (C.c_char * nbytes).from_address(p)   # every view of the new array keeps `buf` alive through .base
            weakref.finalize(buf, L.hpmvs_host_free, p)
            na = np.frombuffer(buf, dtype=a.dtype, count=a.size).reshape(a.shape)
            na[...] = a
            setattr(self, name, na)
        return self

    def c_struct(self) -> PatchBatch:
        b = PatchBatch()
        b.n, b.max_images = self.n, self.max_images
        for name in ("center", "normal", "scale", "n_images", "images", "ok", "color", "ncc", "fmin", "x", "result",
                     "nevals", "stage", "ngrabs"):
            setattr(b, name, _ptr(getattr(self, name)))
        return b


def optimize_batch(scene: Scene, batch: Batch, options: Options | None = None) -> Batch:
    o = options or default_options()
    b = batch.c_struct()
    _chk(lib().hpmvs_optimize_batch(scene.h, C.byref(o), C.byref(b), 0, None))
    return batch


def init_patches_batch(scene: Scene, xyz, meas_off, meas_img, start_level=4, max_images=FAST_IMAGES,
                       options: Options | None = None) -> Batch:
    """Scene::initPatches seed loop on the GPU (reference Scene.cpp:112-178): seeds built on device, refined,
    drift-gated.  Returns the batch (seed patches, refined where ok).  max_images: the rows of the batch's image lists; a point
    with more measurements than that ends at stage 100 (the C ABI's maximum, 64, by default: in a 50-view scene most NVM points
    have more than 32)."""
    o = options or default_options()
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    off = np.ascontiguousarray(meas_off, dtype=np.int32)
    img = np.ascontiguousarray(meas_img, dtype=np.int32)
    n = len(xyz)
    batch = Batch(np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32), np.zeros(n, np.float32),
                  np.zeros(n, np.int32), np.full((n, max_images), -1, np.int32))
    b = batch.c_struct()
    _chk(lib().hpmvs_init_patches_batch(scene.h, C.byref(o), start_level, n, xyz.ctypes.data, off.ctypes.data,
                                        img.ctypes.data if len(img) else None, C.byref(b), 0, None))
    return batch


EXPAND_EXTEND, EXPAND_BRANCH = 0, 1


def expand_directions(mode: int) -> np.ndarray:
    """(cos, sin) pairs of the candidate directions the expansion kernels use; needs no GPU."""
    out = np.zeros(12, dtype=np.float32)
    n = lib().hpmvs_expand_directions(mode, out.ctypes.data)
    if n < 0:
        _chk(n)
    return out[:2 * n].reshape(n, 2)


def expand_batch(scene: Scene, mode: int, parents: Batch, cell_center, cell_width, skip=None,
                 options: Options | None = None) -> Batch:
    """CellProcessor::extend (mode 0, 6 candidates) / ::branch (mode 1, 4 candidates) for a frontier of parent
    cells (reference CellProcessor.cpp:84-178, 210-262): candidates built, optimized and gated on the GPU.
    Returns the batch of n*N candidates (candidate k of parent i at i*N + k)."""
    o = options or default_options()
    N = 6 if mode == EXPAND_EXTEND else 4
    n = parents.n
    cc = np.ascontiguousarray(cell_center, dtype=np.float32).reshape(n, 3)
    cw = np.ascontiguousarray(cell_width, dtype=np.float32).reshape(n)
    sk = None if skip is None else np.ascontiguousarray(skip, dtype=np.uint8).reshape(n * N)
    out = Batch(np.zeros((n * N, 4), np.float32), np.zeros((n * N, 4), np.float32), np.zeros(n * N, np.float32),
                np.zeros(n * N, np.int32), np.full((n * N, parents.max_images), -1, np.int32))
    pb, ob = parents.c_struct(), out.c_struct()
    _chk(lib().hpmvs_expand_batch(scene.h, C.byref(o), mode, C.byref(pb), cc.ctypes.data, cw.ctypes.data,
                                  None if sk is None else sk.ctypes.data, C.byref(ob), 0, None))
    return out


def objective_batch(scene: Scene, batch: Batch, x: np.ndarray, options: Options | None = None):
    o = options or default_options()
    b = batch.c_struct()
    x = np.ascontiguousarray(x, dtype=np.float64)
    f = np.zeros(batch.n, dtype=np.float64)
    g = np.zeros(batch.n, dtype=np.int32)
    _chk(lib().hpmvs_objective_batch(scene.h, C.byref(o), C.byref(b), x.ctypes.data, f.ctypes.data, g.ctypes.data, 0,
                                     None))
    return f, g


def inccs_batch(scene: Scene, batch: Batch, ref_idx=0, robust=0, options: Options | None = None) -> np.ndarray:
    o = options or default_options()
    b = batch.c_struct()
    out = np.zeros((batch.n, batch.max_images), dtype=np.float32)
    _chk(lib().hpmvs_inccs_batch(scene.h, C.byref(o), C.byref(b), ref_idx, robust, out.ctypes.data, 0, None))
    return out


def depth_reset(scene: Scene):
    """Allocate (first call) and clear the scene's depth maps (reference Scene.cpp:74-80)."""
    _chk(lib().hpmvs_scene_depth_reset(scene.h))


def depth_level(scene: Scene, view: int, level: int) -> np.ndarray:
    """One depth map in Eigen's storage order: shape (cols, rows), element [x, y]."""
    r, c = C.c_int(), C.c_int()
    _chk(lib().hpmvs_scene_depth_get_level(scene.h, view, level, None, 0, C.byref(r), C.byref(c)))
    out = np.zeros((c.value, r.value), dtype=np.float32)
    _chk(lib().hpmvs_scene_depth_get_level(scene.h, view, level, out.ctypes.data, out.size, C.byref(r), C.byref(c)))
    return out


def set_depths_batch(scene: Scene, batch: Batch):
    """Scene::setDepths for every patch of the batch with ok != 0 (reference Scene.cpp:351-381)."""
    b = batch.c_struct()
    _chk(lib().hpmvs_set_depths_batch(scene.h, C.byref(b), 0, None))


def depth_ops_batch(scene: Scene, batch: Batch, subtract):
    """Scene::setDepths(patch_i, subtract[i]) for the patches with ok != 0, in batch order, with the sequential loop's maps
    (reference Scene.cpp:351-381; subtract: CellProcessor.cpp:276-279)."""
    b = batch.c_struct()
    sub = np.ascontiguousarray(subtract, dtype=np.uint8).reshape(batch.n)
    _chk(lib().hpmvs_depth_ops_batch(scene.h, C.byref(b), sub.ctypes.data, 0, None))


def level_support_batch(scene: Scene, batch: Batch, min_level=0):
    """Scene::getLevelSupport(patch, min_level) per patch (reference Scene.cpp:334-343)."""
    b = batch.c_struct()
    out = np.zeros(batch.n, dtype=np.int32)
    _chk(lib().hpmvs_level_support_batch(scene.h, C.byref(b), int(min_level), out.ctypes.data, 0, None))
    return out


def depth_gates_batch(scene: Scene, batch: Batch, margin=1.0, abs_int=0):
    """depthTests / viewBlockTest / pixelFreeTests counts per patch (reference Scene.cpp:518-644)."""
    b = batch.c_struct()
    v = np.zeros(batch.n, dtype=np.int32); k = np.zeros(batch.n, dtype=np.int32); f = np.zeros(batch.n, dtype=np.int32)
    _chk(lib().hpmvs_depth_gates_batch(scene.h, C.byref(b), float(margin), int(abs_int), v.ctypes.data, k.ctypes.data,
                                       f.ctypes.data, 0, None))
    return v, k, f


def depth_footprints_batch(scene: Scene, batch: Batch):
    """The depth-map cells the gates of every patch READ and the cells setDepths would WRITE for it (integers; layouts in
    include/hpmvs_amd.h): (writes [n, M, 4], frees [n, M, 4], attached [n, M, 3], view_block [n, n_views, 3])."""
    b = batch.c_struct()
    n, M, V = batch.n, batch.max_images, scene.n_views
    wr = np.zeros((n, M, 4), np.int32); fr = np.zeros((n, M, 4), np.int32)
    at = np.zeros((n, M, 3), np.int32); vb = np.zeros((n, V, 3), np.int32)
    _chk(lib().hpmvs_depth_footprints_batch(scene.h, C.byref(b), wr.ctypes.data, fr.ctypes.data, at.ctypes.data, vb.ctypes.data, 0, None))
    return wr, fr, at, vb


def build_pyramid(img: np.ndarray, device: int = 0) -> np.ndarray:
    h, w, _ = img.shape
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.zeros((h // 2, w // 2, 3), dtype=np.uint8)
    _chk(lib().hpmvs_build_pyramid(device, img.ctypes.data, w, h, out.ctypes.data, 0))
    return out
