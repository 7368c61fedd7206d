/*
 * hpmvs_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's patch-refinement hot path
 * (mo3d::PatchOptimizer::optimize and everything below it), used ONLY by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the
 * checker for the HIP path.  Nothing under hpmvs_amd/ may include, link or
 * call this.
 *
 * Parity status: the photometric part (everything except BOBYQA) is a
 * restatement of reference sources that cannot be compiled here (Eigen3, glog
 * absent) and the reference ships no tests/golden vectors for it -> "parity
 * unpinned" for that part, anchored only by analytic known-answer tests.  The
 * optimiser part is pinned: it runs the reference's genuine vendored NLopt
 * 2.4.2 BOBYQA, compiled in place from /root/reference into oracle/_ref/
 * (see oracle/Makefile); the port (tests/native/bobyqa_port.cpp, its own library) is checked against
 * it evaluation by evaluation.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference root).
 */
#ifndef HPMVS_ORACLE_H
#define HPMVS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 8
#define ORC_MAX_IMAGES 256 /* cap on images attached to one patch (reference: unbounded vector); = HPMVS_MAX_IMAGES */

/* include/hpmvs/HpmvsOptions.h:29-58 (only the fields the path reads) */
typedef struct {
    int MAXLEVEL;              /* 5 */
    int MINLEVEL;              /* 0 */
    float MAX_ANGLE;           /* 60 deg */
    float MIN_ANGLE;           /* 10 deg */
    int MAX_IMAGES_PER_PATCH;  /* 6 (dead in the reference) */
    int MIN_IMAGES_PER_PATCH;  /* 3 */
    float NCC_ALPHA_1;         /* 0.4 */
    float NCC_ALPHA_2;         /* 0.5 */
} orc_options;

/* include/hpmvs/Camera.h:87-105 */
typedef struct {
    int n_levels;
    float P[ORC_MAX_LEVELS][12]; /* row-major 3x4 per level */
    float K0[9];                 /* row-major 3x3, level 0 */
    float center[4];
    float oaxis[4];
    float xaxis[3], yaxis[3], zaxis[3];
} orc_camera;

/* include/hpmvs/Image.h:89-115 storage: interleaved u8 RGB per level */
typedef struct {
    int n_levels;
    int w[ORC_MAX_LEVELS], h[ORC_MAX_LEVELS];
    uint8_t *pix[ORC_MAX_LEVELS];
} orc_image;

typedef struct {
    int n_views;
    orc_camera *cams;
    orc_image *imgs;
    int *covis_off; /* n_views+1 */
    int *covis;     /* flattened lists */
} orc_scene;

/* I/O record of optimize(): include/hpmvs/Patch3d.h:33-83 */
typedef struct {
    float center[4];
    float normal[4];
    float scale;
    int n_images;
    int images[ORC_MAX_IMAGES];
    /* outputs (valid when ok) */
    float ncc_const; /* 1.4f, PatchOptimizer.cpp:95 */
    float color[3];
    /* diagnostics the reference discards */
    double fmin;   /* final mean robust INCC (minf) */
    double x[3];   /* final optimiser variables */
    int nlopt_result;
    int nevals;
    int stage;     /* 0 = ok, else index of the pipeline stage that returned false */
    long ngrabs;   /* sampleTexture calls that passed the gates (588 B each) */
    long ngrabs_all; /* all sampleTexture calls */
} orc_patch;

enum { ORC_OPT_REF = 0, ORC_OPT_PORT = 1 };

void orc_default_options(orc_options *o);

orc_scene *orc_scene_create(int n_views);
void orc_scene_destroy(orc_scene *s);
/* Camera::init (src/hpmvs/Camera.cpp:34-81) + Image::load pyramid (src/hpmvs/Image.cpp:41-66) */
void orc_scene_set_view(orc_scene *s, int i, int w, int h, const uint8_t *rgb_l0,
                        double f, const double q_wxyz[4], const double c[3], int max_level);
void orc_scene_set_covis(orc_scene *s, int i, const int *ids, int n);
const orc_camera *orc_scene_camera(const orc_scene *s, int i);
const uint8_t *orc_scene_level(const orc_scene *s, int i, int level, int *w, int *h);

/* thirdLibs/cimg/CImg.h:21189-21203 on interleaved RGB */
void orc_half_resize(const uint8_t *src, int w, int h, uint8_t *dst);

/* 0 if the chosen optimiser back-end is available */
int orc_optimizer_available(int which);

/* PatchOptimizer::optimize, src/hpmvs/PatchOptimizer.cpp:78-103. returns 1 on success */
int orc_optimize(const orc_scene *s, const orc_options *o, orc_patch *p, int which_optimizer);
void orc_optimize_batch(const orc_scene *s, const orc_options *o, orc_patch *p, int n,
                        int which_optimizer, int n_threads);

/* Scene::initPatches seed loop, src/hpmvs/Scene.cpp:112-178 (without the octree insertion): builds the
 * seed patch of every NVM point, optimizes it and applies the drift gate.  p[i].stage: 0 ok, 1-9 optimize
 * stage, 10 too few measurements, 11 fewer than 2 visible images, 12 drift gate. */
void orc_init_patches(const orc_scene *s, const orc_options *o, int start_level, int n, const double *xyz,
                      const int *meas_off, const int *meas_img, orc_patch *p, int which_optimizer, int n_threads);

/* Candidate loops of CellProcessor::extend (src/hpmvs/CellProcessor.cpp:84-178) and ::branch (:210-262)
 * for one parent patch, up to and including the geometric acceptance gates; the octree look-ups
 * (:120-124 -> `skip`), the depth / view-block / free-pixel tests (:135-142) and the insertion are
 * the caller's.  mode 0 = extend (6 candidates), 1 = branch (4).  out[k].stage: 0 accepted, 1-9/100
 * optimize stage, 20 skipped / outside the cell before optimize, 21 scale gate, 22 drift gate /
 * left the cell.  Returns the fan-out N. */
int orc_expand(const orc_scene *s, const orc_options *o, int mode, const orc_patch *parent,
               const float cell_center[3], float cell_width, const unsigned char *skip, orc_patch *out,
               int which_optimizer);
void orc_expand_batch(const orc_scene *s, const orc_options *o, int mode, int n, const orc_patch *parents,
                      const float *cell_center, const float *cell_width, const unsigned char *skip,
                      orc_patch *out, int which_optimizer, int n_threads);
/* the candidate directions (cosf, sinf of the float angles) */
int orc_expand_directions(int mode, float *dxdy);

/* objective_fn at given optimiser variables x (PatchOptimizer.cpp:286-320) for the patch's
 * current image list; the seed centre/normal define refCenter_/refRay_/axes as optimizePatch does. */
double orc_objective_at(const orc_scene *s, const orc_options *o, const orc_patch *p, const double x[3]);
/* parametersFromCenterNorm for the seed (PatchOptimizer.cpp:416-446) */
void orc_initial_parameters(const orc_scene *s, const orc_options *o, const orc_patch *p, double x[3]);
/* setINCCs (PatchOptimizer.cpp:448-474) */
void orc_inccs(const orc_scene *s, const orc_options *o, const orc_patch *p, int ref_idx, int robust,
               float *out);
/* one sampleTexture (PatchOptimizer.cpp:476-529) with axes from calculatePatchAxis(ref image) */
int orc_sample_texture(const orc_scene *s, const orc_options *o, const orc_patch *p, int ref_slot,
                       int slot, float *tex147, int *level, float *cdxdy6);

/* optimiser-only entry (for pinning the port against the genuine library):
 * minimise f over 3 variables exactly as PatchOptimizer.cpp:348-365 configures NLopt */
typedef double (*orc_obj_fn)(unsigned n, const double *x, double *grad, void *data);
int orc_bobyqa_minimize(int which, int n, orc_obj_fn f, void *data, double *x, const double *lb,
                        const double *ub, double xtol_rel, int maxeval, double *minf, int *nevals);

/* Depth maps (Scene.cpp:74-80: MatrixXf(rows, cols) per view and level, MAX_DEPTH = 1000, column-major) and the three
 * acceptance counts read from them: Scene::setDepths (Scene.cpp:351-381, subtract = false), Scene::depthTests /
 * viewBlockTest / pixelFreeTests (Scene.cpp:518-644).  abs_int selects the reading of the unqualified abs() at
 * Scene.cpp:571 (0: fabsf, 1: C's abs(int)). */
typedef struct orc_depths_s orc_depths;
orc_depths *orc_depths_create(const orc_scene *s);
void orc_depths_destroy(orc_depths *D);
float *orc_depths_level(orc_depths *D, int view, int level, int *rows, int *cols);
void orc_set_depths(const orc_scene *s, orc_depths *D, const orc_patch *p);
/* Scene::setDepths(patch, subtract) (Scene.cpp:351-381; subtract = 1: a cell that holds exactly this patch's depth becomes
 * MAX_DEPTH again), Scene::getLevelSupport (Scene.cpp:334-343), and one priority level of CellProcessor::branch
 * (CellProcessor.cpp:210-307) leaf by leaf: see hpmvs_oracle.c */
void orc_set_depths_ex(const orc_scene *s, orc_depths *D, const orc_patch *p, int subtract);
int orc_level_support(const orc_scene *s, const orc_patch *p, int min_level);
int orc_branch_round(const orc_scene *s, const orc_options *o, orc_depths *D, int n_leaves, const orc_patch *parents,
                     const float *cell_center, const float *cell_width, const unsigned char *final_level, orc_patch *out,
                     unsigned char *split, int which_optimizer);
void orc_depth_gates(const orc_scene *s, const orc_depths *D, const orc_patch *p, float margin, int abs_int,
                     int *n_visible, int *n_blocking, int *n_free);

/* One priority level of the expansion, sequentially (main.cpp:146-181 -> CellProcessor::extend, CellProcessor.cpp:84-178)
 * over a caller-owned occupancy set instead of the octree; frozen_gates: the depth-map counts read the round-start maps
 * (what a frontier-batched round sees).  See hpmvs_oracle.c. */
long long orc_cell_key(const float *p, float width);
int orc_extend_round(const orc_scene *s, const orc_options *o, orc_depths *D, int n_leaves, const orc_patch *parents,
                     float width, float margin, int abs_int, long long *occ_keys, int *n_occ, int occ_cap,
                     int frozen_gates, orc_patch *out, int *counts, int which);

/* test-only switch: evaluation order of Eigen's 4-float reductions (0 left-to-right = default, 1 SSE2, 2 SSE3) */
void orc_set_dot4_order(int order);
/* Eigen's vector / scalar: 0 true division (Eigen >= 3.2, the default and what the HIP path restates), 1 multiply by the
 * reciprocal (Eigen 3.0 / 3.1) */
void orc_set_vec_div_mode(int mode);
/* diagnosis only: bit 0 correctly rounded acos, bit 1 the device's sin / cos series (hpmvs_oracle.c) */
void orc_set_libm_mode(int bits);
double orc_acos_in_mode(double u, int bits); /* the acos() the path would call under `bits` (tests) */

#ifdef __cplusplus
}
#endif

#endif
