/*
 * hpmvs_amd.h -- C ABI of the MI355X-native HPMVS patch-refinement path.
 *
 * This is the drop-in boundary for the ONE hot path of alexlocher/hpmvs:
 * mo3d::PatchOptimizer::optimize(Patch3d&) and everything below it
 * (reference src/hpmvs/PatchOptimizer.cpp:78-103).  Plain pointers and sizes only;
 * no C++/torch types.  The C++ mirror of the reference interface
 * (include/hpmvs/PatchOptimizer.h, Patch3d.h, ...) is a thin layer over these calls.
 *
 * Every entry point returns HPMVS_OK or a negative error; hpmvs_last_error() gives text.
 * Per-patch failure ("drop this patch", the reference's `return false`) is NOT an error:
 * it is reported in hpmvs_patch_batch::ok.
 *
 * Reference interfaces replaced (file:line relative to the reference root):
 *   hpmvs_scene_*            <- the const Scene view PatchOptimizer caches at construction,
 *                               src/hpmvs/PatchOptimizer.cpp:38-41 (cameras_, images_, covis_;
 *                               include/hpmvs/Scene.h:69-71), filled by Scene::addCameras
 *                               (src/hpmvs/Scene.cpp:42-88: Image::load pyramid, Camera::init)
 *                               and Scene::extractCoVisiblilty (Scene.cpp:241-298).
 *   hpmvs_optimize_batch     <- PatchOptimizer::optimize, PatchOptimizer.cpp:78-103, over a batch
 *                               (the OpenMP seed loop of Scene::initPatches, Scene.cpp:114-167).
 *   hpmvs_objective_batch    <- PatchOptimizer::static_objective_fn / objective_fn,
 *                               PatchOptimizer.cpp:286-320 (what NLopt calls back).
 *   hpmvs_inccs_batch        <- PatchOptimizer::setINCCs, PatchOptimizer.cpp:448-474.
 *   hpmvs_build_pyramid      <- Image::load's pyramid, src/hpmvs/Image.cpp:55-63
 *                               (CImg get_resize_halfXY, thirdLibs/cimg/CImg.h:21189-21203).
 *   hpmvs_init_patches_batch <- the seed loop of Scene::initPatches, src/hpmvs/Scene.cpp:112-178.
 *   hpmvs_expand_batch       <- the candidate loops of CellProcessor::extend / ::branch,
 *                               src/hpmvs/CellProcessor.cpp:84-142 and :210-262.
 *   hpmvs_camera_from_nvm    <- Camera::init, src/hpmvs/Camera.cpp:34-81.
 */
#ifndef HPMVS_AMD_H
#define HPMVS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HPMVS_OK 0
#define HPMVS_ERR_HIP (-1)     /* a HIP runtime call failed */
#define HPMVS_ERR_ARG (-2)     /* bad argument */
#define HPMVS_ERR_STATE (-3)   /* scene not committed / already committed */
#define HPMVS_ERR_NODEVICE (-4) /* no gfx950 device visible: there is NO CPU fallback */

#define HPMVS_MAX_LEVELS 8
#define HPMVS_MAX_IMAGES 256 /* images attached to one patch (reference: unbounded vector<int>); overflow => stage 100.  Lists of up to
                              HPMVS_FAST_IMAGES ids run in the batch kernel; a patch whose list outgrows that at any point of the
                              pipeline is redone by the wide kernel behind it (same launch, same results, one patch per wavefront) */
#define HPMVS_FAST_IMAGES 64
#define HPMVS_RECORD_IMAGES 64 /* ids an hpmvs_record carries */

typedef struct hpmvs_scene hpmvs_scene; /* opaque; owns the HBM-resident pyramids and tables */

/* include/hpmvs/HpmvsOptions.h:29-58, the fields the path reads */
typedef struct {
    int32_t MAXLEVEL;             /* 5 */
    int32_t MINLEVEL;             /* 0 */
    float MAX_ANGLE;              /* 60 deg in rad */
    float MIN_ANGLE;              /* 10 deg in rad */
    int32_t MAX_IMAGES_PER_PATCH; /* 6 (dead in the reference) */
    int32_t MIN_IMAGES_PER_PATCH; /* 3 */
    float NCC_ALPHA_1;            /* 0.4 */
    float NCC_ALPHA_2;            /* 0.5 */
} hpmvs_options;

/* What the path reads of mo3d::Camera (include/hpmvs/Camera.h:87-105).  Level l's projection is
 * diag(2^-l, 2^-l, 1) * P0 exactly (src/hpmvs/Camera.cpp:55-63 halves rows 0,1 per level), so only
 * level 0 is passed. */
typedef struct {
    float P0[12];    /* projection_[0], row-major 3x4 */
    float center[4]; /* center_ (w = 1) */
    float xaxis[3];  /* xAxis_ */
    float yaxis[3];  /* yAxis_ */
    float zaxis[3];  /* zAxis_ */
    float fsum;      /* kMat_[0](0,0) + kMat_[0](1,1) */
    int32_t n_levels; /* projection_.size() = MAXLEVEL + 1 */
} hpmvs_camera;

/* A batch of Patch3d records (include/hpmvs/Patch3d.h:33-83) in structure-of-arrays form.
 * All pointers are host pointers unless `on_device` is set in the call, in which case all are
 * device pointers on the scene's GPU.  Inputs are updated in place only for patches with ok=1
 * (reference: patch untouched on failure, PatchOptimizer.cpp:86-87). */
typedef struct {
    int32_t n;           /* number of patches */
    int32_t max_images;  /* row stride of `images` (<= HPMVS_MAX_IMAGES) */
    float *center;       /* [n][4] center_ (w = 1)            in/out */
    float *normal;       /* [n][4] normal_ (w = 0)            in/out */
    float *scale;        /* [n]    scale_3dx_                 in (returned unchanged) */
    int32_t *n_images;   /* [n]    images_.size()             in/out */
    int32_t *images;     /* [n][max_images] images_, [0] = reference image   in/out */
    uint8_t *ok;         /* [n]    return value of optimize() out */
    float *color;        /* [n][3] color_ (valid when ok)     out */
    float *ncc;          /* [n]    ncc_ (constant 1.4f as in the reference, PatchOptimizer.cpp:95) out */
    /* diagnostics the reference computes and discards (may be NULL) */
    double *fmin;        /* [n]    final mean robust INCC (NLopt minf) */
    double *x;           /* [n][3] final optimiser variables (depth, angle1, angle2) */
    int32_t *result;     /* [n]    nlopt_result code of the BOBYQA run */
    int32_t *nevals;     /* [n]    objective evaluations */
    int32_t *stage;      /* [n]    0 = ok, else index of the pipeline stage that returned false (1..9);
                          *        100 = attached-image list overflow, 101 = an image id outside the scene */
    int32_t *ngrabs;     /* [n]    sampleTexture calls that passed the gates (588 B of image each) */
} hpmvs_patch_batch;

/* ---- library ---------------------------------------------------------------------------- */
const char *hpmvs_last_error(void);
int hpmvs_device_count(void);
/* Identifies the build of this library: a hash of its sources taken by the Makefile.  Measurement records
 * (profiles/pmc_traffic.json) carry it, so that counters are never reported beside another kernel's time. */
const char *hpmvs_build_id(void);
void hpmvs_default_options(hpmvs_options *o);

/* Host-side Camera::init (reference src/hpmvs/Camera.cpp:34-81): NVM camera (focal length,
 * rotation quaternion wxyz world->camera in double, centre; include/hpmvs/NVMReader.h:44-50) and the
 * level-0 image size -> the float32 tables above.  Pure host code, usable without a GPU. */
int hpmvs_camera_from_nvm(double f, const double q_wxyz[4], const double c[3], int width, int height,
                          int max_level, hpmvs_camera *out);

/* ---- scene (HBM-resident, immutable after commit; shared read-only by all callers) ------- */
int hpmvs_scene_create(int n_views, int device, hpmvs_scene **out);
/* Level-0 interleaved u8 RGB (row-major, 3*(y*W+x)+c: reference Image.h:93-105).  The pyramid is
 * built on the GPU by the half-resize kernel.  rgb_on_device != 0: `rgb_l0` is a device pointer. */
int hpmvs_scene_set_view(hpmvs_scene *s, int view, int width, int height, const uint8_t *rgb_l0,
                         int rgb_on_device, const hpmvs_camera *cam);
int hpmvs_scene_set_covis(hpmvs_scene *s, int view, const int32_t *ids, int n);
int hpmvs_scene_commit(hpmvs_scene *s);
int hpmvs_scene_destroy(hpmvs_scene *s);
/* copy one pyramid level back to the host (tests: bit-exact pyramid parity) */
int hpmvs_scene_get_level(const hpmvs_scene *s, int view, int level, uint8_t *host_out, size_t cap,
                          int *w, int *h);
size_t hpmvs_scene_bytes(const hpmvs_scene *s);

/* stand-alone pyramid kernel: src (w x h, device or host) -> dst (w/2 x h/2) */
int hpmvs_build_pyramid(int device, const uint8_t *src, int w, int h, uint8_t *dst, int on_device);

/* ---- the hot path ------------------------------------------------------------------------- */
/* Full optimize() for every patch of the batch.  `stream` is a hipStream_t (NULL = default
 * stream); with on_device != 0 the call only enqueues work on `stream`. */
int hpmvs_optimize_batch(const hpmvs_scene *s, const hpmvs_options *o, hpmvs_patch_batch *b,
                         int on_device, void *stream);

/* The seed-initialisation loop of Scene::initPatches (reference src/hpmvs/Scene.cpp:112-178) as one
 * call: for every NVM point build the seed patch on the GPU (centre = xyz, attached images = the
 * point's measurements that project inside the START_LEVEL image with a 2 px margin, normal towards the
 * first attached camera, scale = getScale(centre, START_LEVEL)), run optimize() on the whole batch,
 * and apply the post-gate |centre - xyz| <= 2 * scale (Scene.cpp:171).
 *   xyz[n][3] (float64), meas_off[n+1], meas_img[meas_off[n]] : NVM_Point::xyz / measurements[].imgIndex
 * `b` receives the seed patches (refined where ok[i] = 1).  stage[i] additionally uses
 *   10 = fewer than MIN_IMAGES_PER_PATCH measurements (Scene.cpp:127), 11 = fewer than 2 visible images
 *   (Scene.cpp:153), 12 = failed the drift gate (Scene.cpp:171).
 * Host or device pointers as for hpmvs_optimize_batch. */
int hpmvs_init_patches_batch(const hpmvs_scene *s, const hpmvs_options *o, int start_level, int n_points,
                             const double *xyz, const int32_t *meas_off, const int32_t *meas_img,
                             hpmvs_patch_batch *b, int on_device, void *stream);

/* Frontier expansion: the candidate loops of CellProcessor::extend (reference
 * src/hpmvs/CellProcessor.cpp:84-178) and CellProcessor::branch (:210-262) for a whole frontier of
 * octree cells in one call -- candidate construction, optimize() and the geometric acceptance gates.
 *   mode HPMVS_EXPAND_EXTEND : 6 candidates per parent on a hexagon of radius cell_width around the parent
 *        centre in the parent's tangent plane (x axis from the reference camera), scale = 0.9 * width / 2;
 *        accepted if optimize() succeeds, width/2 < 2*scale < width (:131-132) and the centre moved less
 *        than 1.5 * width from the PARENT centre (:133).
 *   mode HPMVS_EXPAND_BRANCH : 4 candidates at radius width/4 on the diagonals, scale = 0.45 * width / 2;
 *        only candidates inside the parent's cell are optimized (:247) and they must still be inside it
 *        afterwards (:257).
 * parents: n patches (center, normal, n_images, images are read; images[.][0] is the reference image);
 * cell_center[n][3], cell_width[n]: the octree leaf of each parent (Cell::c_, Cell::width_);
 * skip (optional, n*N bytes): nonzero = the caller already knows the candidate is not wanted (extend: the
 *   target leaf is occupied or finer, :120-124) -> it is built but not optimized;
 * out: batch of n*N patches, candidate k of parent i at index i*N + k, max_images = parents->max_images.
 *   ok = accepted; stage = 0, the optimize() stage 1..9/100, 20 = skipped / outside the cell before optimize,
 *   21 = scale gate, 22 = drift gate (extend) or left the cell (branch).  A candidate that fails keeps its
 *   constructed centre/scale (optimize() leaves failed patches untouched).
 * The depth / view-block / free-pixel tests (:135-142) and the octree insertion read and write scene state
 * that is order dependent in the reference; they stay with the caller.  Host or device pointers as for
 * hpmvs_optimize_batch. */
#define HPMVS_EXPAND_EXTEND 0
#define HPMVS_EXPAND_BRANCH 1
int hpmvs_expand_batch(const hpmvs_scene *s, const hpmvs_options *o, int mode, const hpmvs_patch_batch *parents,
                       const float *cell_center, const float *cell_width, const uint8_t *skip,
                       hpmvs_patch_batch *out, int on_device, void *stream);
/* the (cos, sin) pairs of the candidate directions the kernels use: dxdy[2*N], N = 6 (extend) or 4 (branch);
 * returns N.  (std::cos/std::sin of the float angle 2*pi/N*i [+ pi/4], CellProcessor.cpp:107-109, 233-235.) */
int hpmvs_expand_directions(int mode, float *dxdy);

/* objective_fn at optimiser variables x[n][3] for each patch's current image list, with
 * refCenter_/refRay_/axes taken from the patch as optimizePatch does.  f_out[n]. */
int hpmvs_objective_batch(const hpmvs_scene *s, const hpmvs_options *o, const hpmvs_patch_batch *b,
                          const double *x, double *f_out, int32_t *ngrabs_out, int on_device,
                          void *stream);

/* setINCCs(ref_idx, robust) for each patch: out[n][max_images] */
int hpmvs_inccs_batch(const hpmvs_scene *s, const hpmvs_options *o, const hpmvs_patch_batch *b,
                      int ref_idx, int robust, float *out, int on_device, void *stream);

/* ---- depth maps and the acceptance gates of the expansion (second half) ----------------------------------
 * The reference accepts a refined candidate only if it agrees with the depth maps written by the patches accepted
 * so far (src/hpmvs/CellProcessor.cpp:134-142, 198-200):
 *     Scene::depthTests(p, margin)    >= MIN_IMAGES_PER_PATCH      src/hpmvs/Scene.cpp:518-524, 531-580
 *     Scene::viewBlockTest(p, margin) <  MIN_IMAGES_PER_PATCH      src/hpmvs/Scene.cpp:607-642 (every view of the scene)
 *     Scene::pixelFreeTests(p)        >= MIN_IMAGES_PER_PATCH - 1 and > 0.75 * #images    src/hpmvs/Scene.cpp:582-605
 * and records an accepted patch with Scene::setDepths (src/hpmvs/Scene.cpp:351-381).  The maps live in HBM in the
 * reference's own layout -- per view and pyramid level an Eigen::MatrixXf(rows, cols), column-major, filled with
 * MAX_DEPTH = 1000, rows / cols = level size / DEPTH_SUBSAMPLE (src/hpmvs/Scene.cpp:33,74-80) -- so the unchanged
 * scheduler copies its matrices in and out as they are (matrix.data(), rows(), cols()).
 *   hpmvs_scene_depth_reset      allocate (first call) and fill every map with MAX_DEPTH
 *   hpmvs_scene_depth_set_level / _get_level   one map <-> host (get with data == NULL: shape query)
 *   hpmvs_set_depths_batch       setDepths(patch, subtract = false) for every patch of the batch with ok[i] != 0
 *                                (all patches when ok is NULL): a float minimum per cell, order-independent
 *   hpmvs_depth_gates_batch      the three counts per patch, read-only; the caller applies the thresholds.
 *                                abs_int selects the reading of the unqualified `abs(diff)` at Scene.cpp:571: 0 = the
 *                                <cmath> overload (fabsf), 1 = C's abs(int), which truncates the difference first --
 *                                which one a reference binary has depends on the headers its toolchain exports. */
int hpmvs_scene_depth_reset(hpmvs_scene *s);
int hpmvs_scene_depth_set_level(hpmvs_scene *s, int view, int level, const float *data, int rows, int cols);
int hpmvs_scene_depth_get_level(const hpmvs_scene *s, int view, int level, float *data, size_t capacity, int *rows, int *cols);
int hpmvs_set_depths_batch(hpmvs_scene *s, const hpmvs_patch_batch *b, int on_device, void *stream);
int hpmvs_depth_gates_batch(const hpmvs_scene *s, const hpmvs_patch_batch *b, float margin, int abs_int,
                            int32_t *n_visible, int32_t *n_blocking, int32_t *n_free, int on_device, void *stream);
/* Scene::setDepths(patch_i, subtract[i]) for i = 0 .. n - 1 with ok[i] != 0, IN THAT ORDER (src/hpmvs/Scene.cpp:351-381).
 * subtract[i] != 0 takes patch i's depths back: a cell that still holds exactly that patch's depth becomes MAX_DEPTH again
 * (:373-374) -- what CellProcessor::branch does for the patch of a leaf it splits before it enters the children's depths
 * (src/hpmvs/CellProcessor.cpp:276-279, 296).  Unlike the minimum of subtract = 0 that depends on the order of the calls that
 * reach a cell, so the batch is applied cell by cell in call order (sorted keys, one thread replays a cell's calls): the maps
 * are those of the sequential loop.  subtract == NULL: hpmvs_set_depths_batch.  n * max_images < 2^28 per call.  The call is
 * host-synchronous also with on_device = 1 (the number of keys is read back before the sort; its temporaries are freed on return).
 * hpmvs_level_support_batch: Scene::getLevelSupport(patch, min_level) (src/hpmvs/Scene.cpp:334-343), branch's first gate. */
int hpmvs_depth_ops_batch(hpmvs_scene *s, const hpmvs_patch_batch *b, const uint8_t *subtract, int on_device, void *stream);
int hpmvs_level_support_batch(const hpmvs_scene *s, const hpmvs_patch_batch *b, int min_level, int32_t *support, int on_device, void *stream);
/* The cells those calls touch, as integers: what a scheduler needs to run one priority level's candidates in conflict-free
 * waves and still end with the reference's SEQUENTIAL result (CellProcessor.cpp:130-142 reads maps that earlier candidates
 * of the same level have written, Scene.cpp:351-381; hpmvs_amd/frontier.py is that scheduler, INTEGRATION.md has the recipe).
 * Any output may be NULL.  Computed with the same device code as the gates / setDepths, so the cells are exact.
 *   writes     [n][max_images][4]  attached image k: view, level, x, y of the cell setDepths offers its depth to (view -1: none)
 *   frees      [n][max_images][4]  attached image k: view, level, x, y of the cell pixelFreeTest reads (view -1: none)
 *   attached   [n][max_images][3]  attached image k: view, ix0, iy0 -- depthTests reads getFullDepth of the 3x3 level-0 pixel
 *                                  block from (ix0, iy0): cell ((int)(px / 2.0) >> l, (int)(py / 2.0) >> l) of every level l
 *   view_block [n][n_views][3]     every view v: 1 if viewBlockTest examines it, ix0, iy0 of its 3x3 block (read the same way) */
int hpmvs_depth_footprints_batch(const hpmvs_scene *s, const hpmvs_patch_batch *b, int32_t *writes, int32_t *frees,
                                 int32_t *attached, int32_t *view_block, int on_device, void *stream);

/* Host-pointer calls and pinned memory.  An array of a host-pointer call (on_device = 0) that lies in pinned host memory
 * mapped into the GPU's address space -- hipHostMalloc / hipHostRegister, torch's pin_memory(), hpmvs_host_alloc below --
 * is used IN PLACE: the kernels read a patch's inputs once and write its outputs once, so they travel over PCIe while the
 * kernel runs; pageable arrays are copied through device buffers as before, array by array.  Results are identical either
 * way (outputs the kernel does not write for a failed patch -- color, ncc -- read 0, as after a staged call).  The
 * reference has no counterpart: its patches live in host containers (Scene.cpp:94-96, CellProcessor.cpp:129).
 * HPMVS_ZERO_COPY=0 in the environment switches the detection off.
 * ON AN INFRASTRUCTURE ERROR (negative status: a HIP error while enqueueing, no workspace) the two forms differ: a staged
 * call leaves the caller's arrays untouched, an in-place call may already have refined some patches in place (center,
 * normal, n_images, images) and zero-filled color / ncc -- treat the in/out arrays of a pinned batch as undefined after a
 * negative status and rebuild them.  Argument errors (HPMVS_ERR_ARG / _STATE) are detected before the first write in
 * both forms.  Per-patch failure (ok[i] == 0) is not an error and leaves that patch untouched in both forms. */
void *hpmvs_host_alloc(size_t bytes);   /* NULL on failure */
void hpmvs_host_free(void *p);
/* of the last host-pointer hpmvs_optimize_batch call above the small-batch thresholds: bytes copied through device
 * buffers and bytes used in place */
int hpmvs_last_staging(const hpmvs_scene *s, unsigned long long *staged_bytes, unsigned long long *in_place_bytes);

/* last optimize launch: kernel time measured with HIP events on the launch stream (ms).  Meaningful with ONE caller at a
 * time: the scene holds a single event pair, so concurrent launches from several host threads can pair the start of one
 * launch with the end of another. */
int hpmvs_last_kernel_ms(const hpmvs_scene *s, float *ms);
/* How many patches of the last refinement launch outgrew the batch kernel's HPMVS_FAST_IMAGES-id rows and were redone by the
 * wide kernel behind it (synchronises the device). */
int hpmvs_last_wide_patches(const hpmvs_scene *s, int32_t *n);

/* ---- multi-GPU: the per-round exchange of refined patches ---------------------------------------------------------
 * Patches are independent (PatchOptimizer.cpp:78-103 reads only its patch and the immutable scene; the reference
 * runs them as an OpenMP loop, Scene.cpp:94-96,114), so every rank (one process per GPU, scene replicated)
 * refines a contiguous slice of the round's batch and the only exchange is ONE all-gather of fixed-size records,
 * after which every rank holds the whole refined set for its host scheduler (main.cpp:145-181: one barrier per
 * priority level).  These entry points make that reachable from a C/C++ host without Python:
 *   hpmvs_pack_records       SoA batch (device arrays) -> n records (device), one kernel on `stream`
 *   hpmvs_allgather_records  ncclAllGather of `count` records per rank over the caller's RCCL communicator
 *                            (xGMI inside a node); RCCL is looked up at run time in the process (the library the
 *                            communicator came from), else librccl.so.1 is opened -- no link-time dependency;
 *                            $HPMVS_RCCL_LIBRARY names another library to take it from.  HPMVS_ERR_STATE when
 *                            none is found; HPMVS_ERR_ARG for a null communicator, for host pointers and for
 *                            send / recv buffers on two different devices (checked before the collective starts)
 *   hpmvs_unpack_records     records (device) -> SoA batch (device arrays); ok / color / fmin included
 * Ragged rounds (n not divisible by the ranks): pad every rank to the largest slice, records beyond a rank's
 * count are zero (ok = 0). */
typedef struct hpmvs_record {
    float center[4];
    float normal[4];
    float color[3];
    float scale;
    double fmin;
    uint8_t ok;
    uint8_t pad0;
    uint16_t n_images;
    uint8_t pad1[4];
    uint16_t images[HPMVS_RECORD_IMAGES]; /* unused slots 0xFFFF; n_images > HPMVS_RECORD_IMAGES: the first 64 ids */
} hpmvs_record;                        /* 192 bytes */
/* Lists longer than the record's HPMVS_RECORD_IMAGES ids (rare: a refined list of a dense scene): the record keeps the true
 * n_images and the first 64 ids, the ids from 64 on travel in a TAIL of the sender's overflow segment -- one more all-gather,
 * made only when some rank has a tail (every rank can tell from the gathered records: n_images > 64 with ok set).  The 192-byte
 * record is unchanged (format version 1 = no tails anywhere = what rounds 1-5 sent).
 *   hpmvs_pack_record_tails    tails of the refined patches (ok != 0) with n_images > 64, in patch order; *n_tails (host int) is
 *                              their number (the call synchronises the stream); HPMVS_ERR_ARG when `cap` tails do not hold them
 *   hpmvs_unpack_record_tails  ids 64.. of patch (patch_offset + tail.patch) of a batch that hpmvs_unpack_records has filled */
typedef struct hpmvs_record_tail {
    int32_t patch;   /* index of the patch in the sender's shard */
    uint16_t count;  /* ids in this tail = n_images - 64 */
    uint16_t pad;
    uint16_t images[HPMVS_MAX_IMAGES - HPMVS_RECORD_IMAGES];
} hpmvs_record_tail; /* 392 bytes */
int hpmvs_pack_record_tails(const hpmvs_patch_batch *b, hpmvs_record_tail *tails, int32_t cap, int32_t *n_tails, void *stream);
int hpmvs_unpack_record_tails(const hpmvs_record_tail *tails, int32_t n_tails, int32_t patch_offset, hpmvs_patch_batch *b, void *stream);
int hpmvs_pack_records(const hpmvs_patch_batch *b, hpmvs_record *records, void *stream);
int hpmvs_unpack_records(const hpmvs_record *records, int n, hpmvs_patch_batch *b, void *stream);
int hpmvs_allgather_records(void *nccl_comm, const hpmvs_record *send, size_t count, hpmvs_record *recv, void *stream);

/* ---- diagnostics --------------------------------------------------------------------------------------------------
 * The optimiser that optimize_kernel runs per lane (NLopt 2.4.2 LN_BOBYQA as PatchOptimizer.cpp:348-365 configures
 * it: 3 variables, xtol_rel 1e-7, default initial step), driven ON THE GPU by the analytic objectives of
 * hpmvs_amd/csrc/selftest_obj.h instead of the photometric one, 32 different problems per wavefront.  Exists so
 * that the device build of the blocks the photometric objective never reaches (rescue_, bobyqa.c:143-742; active
 * bounds; ROUNDOFF_LIMITED) can be compared evaluation by evaluation with the genuine library.  Host arrays:
 * kind[n], params[n][8], x0/lb/ub[n][3] in; xfinal[n][3], minf[n], result[n] (nlopt_result), nevals[n],
 * rescue_calls[n] out; trace[n][trace_cap][4] = (x0, x1, x2, f) of every evaluation (may be NULL). */
int hpmvs_selftest_bobyqa(int device, int n, const int32_t *kind, const double *params, const double *x0,
                          const double *lb, const double *ub, int maxeval, double *xfinal, double *minf,
                          int32_t *result, int32_t *nevals, int32_t *rescue_calls, double *trace, int trace_cap);

#ifdef __cplusplus
}
#endif
#endif /* HPMVS_AMD_H */
