/*
 * hpmvs_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See hpmvs_oracle.h for the rules and the parity status.
 *
 * Plain-C restatement of the reference hot path.  All geometry/texture
 * arithmetic is float32 in the reference's evaluation order (sequential sums,
 * no FMA contraction: build with -ffp-contract=off); optimiser variables are
 * float64.  File:line citations are relative to the reference root.
 */
#define _GNU_SOURCE
#include "hpmvs_oracle.h"

#include <dlfcn.h>
#include <limits.h>
#include <math.h>
#include <quadmath.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------- */
/* tiny float vector helpers (Eigen fixed-size ops restated as scalar loops)  */

static inline float dot3(const float *a, const float *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
/* Order of Eigen's 4-float reductions (Vector4f::dot / norm / normalized, the rows of P * X).  Which one an
 * hpmvs binary has depends on the SIMD ISA Eigen was compiled for, and Eigen itself is absent from this image:
 *   0  left to right            ((x0+x1)+x2)+x3   no vectorisation              <- what the HIP path restates
 *   1  SSE2 packet reduction    (x0+x2)+(x1+x3)   predux: movehl + add, shuffle + add
 *   2  SSE3 horizontal add      (x0+x1)+(x2+x3)   haddps twice
 * tests/test_cpu_oracle.py::test_eigen_reduction_order_sensitivity runs the oracle under all three and reports
 * how far the refined patches move (the "stated float tolerance" against a real build). */
static int g_dot4_order = 0;
void orc_set_dot4_order(int order) { g_dot4_order = order; }
/* A second build-dependent choice inside Eigen: `vector / scalar` (hence normalized(), `ave /= size3` in
 * Patch2d.hpp:60 and `result /= result(2)` in Camera.h:55) is a true division per coefficient from Eigen 3.2 on,
 * but a multiplication by the reciprocal (scalar_quotient1_op: m_other = 1 / other) in Eigen 3.0 / 3.1.
 *   0  true division   <- what the HIP path restates      1  multiply by 1/scalar */
static int g_vec_div_mode = 0;
void orc_set_vec_div_mode(int mode) { g_vec_div_mode = mode; }
/* A third one, for DIAGNOSIS only (tools/full_parity.py): which libm the float64 transcendental calls of the path see.
 * The reference calls the C library's acos / sin / cos (PatchOptimizer.cpp:405-407, 429-437); glibc 2.35's are < 1 ulp but not
 * correctly rounded, the device cannot call them, and BOBYQA occasionally amplifies a last-bit difference of the start point
 * or of a normal.  With these bits the oracle uses what the DEVICE uses, so that a differing patch can be attributed:
 *   bit 0  acos correctly rounded (libquadmath's acosq rounded to double -- x87's acosl is not enough: for u = 0.9677955508232117 it
 *          rounds to glibc's value, one ulp above the correctly rounded one; the device: hpmvs_amd/csrc/cr_acos.hpp)
 *   bit 1  sin / cos of setCenterNorm by the device's Horner series (photometric.hpp sincos_halfpi, same IEEE operations)
 *   0      the host libm  <- every parity statement of the repo */
static int g_libm_mode = 0;
void orc_set_libm_mode(int bits) { g_libm_mode = bits; }
static double path_acos(double u) { return (g_libm_mode & 1) ? (double)acosq((__float128)u) : acos(u); }
double orc_acos_in_mode(double u, int bits) { const int keep = g_libm_mode; g_libm_mode = bits; const double r = path_acos(u); g_libm_mode = keep; return r; }  /* (tests) */
static void path_sincos(double a, double *s, double *c) {
    if (!(g_libm_mode & 2)) { *s = sin(a); *c = cos(a); return; }
    const double z = a * a;
    double ps = -1.0 / 51090942171709440000.0;
    ps = ps * z + 1.0 / 121645100408832000.0;
    ps = ps * z - 1.0 / 355687428096000.0;
    ps = ps * z + 1.0 / 1307674368000.0;
    ps = ps * z - 1.0 / 6227020800.0;
    ps = ps * z + 1.0 / 39916800.0;
    ps = ps * z - 1.0 / 362880.0;
    ps = ps * z + 1.0 / 5040.0;
    ps = ps * z - 1.0 / 120.0;
    ps = ps * z + 1.0 / 6.0;
    *s = a - a * z * ps;
    double pc = 1.0 / 1124000727777607680000.0;
    pc = pc * z - 1.0 / 2432902008176640000.0;
    pc = pc * z + 1.0 / 6402373705728000.0;
    pc = pc * z - 1.0 / 20922789888000.0;
    pc = pc * z + 1.0 / 87178291200.0;
    pc = pc * z - 1.0 / 479001600.0;
    pc = pc * z + 1.0 / 3628800.0;
    pc = pc * z - 1.0 / 40320.0;
    pc = pc * z + 1.0 / 720.0;
    pc = pc * z - 1.0 / 24.0;
    pc = pc * z + 0.5;
    *c = 1.0 - z * pc;
}
static inline float vdiv(float a, float s, float inv_s) { return g_vec_div_mode ? a * inv_s : a / s; }
static inline float sum4(float p0, float p1, float p2, float p3) {
    if (g_dot4_order == 1) return (p0 + p2) + (p1 + p3);
    if (g_dot4_order == 2) return (p0 + p1) + (p2 + p3);
    return ((p0 + p1) + p2) + p3;
}
static inline float dot4(const float *a, const float *b) {
    return sum4(a[0] * b[0], a[1] * b[1], a[2] * b[2], a[3] * b[3]);
}
static inline void cross3(const float *a, const float *b, float *r) {
    const float r0 = a[1] * b[2] - a[2] * b[1];
    const float r1 = a[2] * b[0] - a[0] * b[2];
    const float r2 = a[0] * b[1] - a[1] * b[0];
    r[0] = r0; r[1] = r1; r[2] = r2;
}
static inline void normalized3(const float *a, float *r) {
    const float n2 = dot3(a, a);
    if (n2 > 0.0f) { const float n = sqrtf(n2), in = 1.0f / n; r[0] = vdiv(a[0], n, in); r[1] = vdiv(a[1], n, in); r[2] = vdiv(a[2], n, in); }
    else { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
}
static inline void normalized4(const float *a, float *r) {
    const float n2 = dot4(a, a);
    if (n2 > 0.0f) { const float n = sqrtf(n2), in = 1.0f / n; for (int i = 0; i < 4; i++) r[i] = vdiv(a[i], n, in); }
    else { for (int i = 0; i < 4; i++) r[i] = a[i]; }
}
static inline void sub4(const float *a, const float *b, float *r) { for (int i = 0; i < 4; i++) r[i] = a[i] - b[i]; }
static inline void add4(const float *a, const float *b, float *r) { for (int i = 0; i < 4; i++) r[i] = a[i] + b[i]; }
static inline float norm4(const float *a) { return sqrtf(dot4(a, a)); }

/* ------------------------------------------------------------------------- */

void orc_default_options(orc_options *o) {
    /* include/hpmvs/HpmvsOptions.h:29-58 */
    o->MAXLEVEL = 5;
    o->MINLEVEL = 0;
    o->MAX_ANGLE = (float)(60.0f * M_PI / 180.0f);
    o->MIN_ANGLE = (float)(10.0f * M_PI / 180.0f);
    o->MAX_IMAGES_PER_PATCH = 6;
    o->MIN_IMAGES_PER_PATCH = 3;
    o->NCC_ALPHA_1 = 0.4f;
    o->NCC_ALPHA_2 = 0.5f;
}

orc_scene *orc_scene_create(int n_views) {
    orc_scene *s = (orc_scene *)calloc(1, sizeof(orc_scene));
    s->n_views = n_views;
    s->cams = (orc_camera *)calloc((size_t)n_views, sizeof(orc_camera));
    s->imgs = (orc_image *)calloc((size_t)n_views, sizeof(orc_image));
    s->covis_off = (int *)calloc((size_t)n_views + 1, sizeof(int));
    s->covis = NULL;
    return s;
}

void orc_scene_destroy(orc_scene *s) {
    if (!s) return;
    for (int i = 0; i < s->n_views; i++)
        for (int l = 0; l < ORC_MAX_LEVELS; l++) free(s->imgs[i].pix[l]);
    free(s->cams); free(s->imgs); free(s->covis_off); free(s->covis); free(s);
}

/* thirdLibs/cimg/CImg.h:21189-21203 (get_resize_halfXY) with the 3x3 neighbourhood walk of
 * cimg_for3x3 (CImg.h:1159-1175: previous/next indices clamped to the image) applied per
 * channel; the reference runs it on planar data before permute_axes("cxyz")
 * (src/hpmvs/Image.cpp:55-63), which is the same arithmetic per channel. */
void orc_half_resize(const uint8_t *src, int w, int h, uint8_t *dst) {
    const float m0 = 0.07842776544f, m1 = 0.1231940459f, m4 = 0.1935127547f;
    const int w2 = w / 2, h2 = h / 2;
    for (int c = 0; c < 3; c++)
        for (int y = 1; y < h; y += 2) {
            const int yp = y - 1, yn = (y + 1 < h) ? y + 1 : h - 1;
            if (y / 2 >= h2) break;
            for (int x = 1; x < w; x += 2) {
                const int xp = x - 1, xn = (x + 1 < w) ? x + 1 : w - 1;
                if (x / 2 >= w2) break;
#define PX(xx, yy) ((float)src[3 * ((size_t)(yy) * w + (xx)) + c])
                const float v = PX(xp, yp) * m0 + PX(x, yp) * m1 + PX(xn, yp) * m0 +
                                PX(xp, y) * m1 + PX(x, y) * m4 + PX(xn, y) * m1 +
                                PX(xp, yn) * m0 + PX(x, yn) * m1 + PX(xn, yn) * m0;
#undef PX
                dst[3 * ((size_t)(y / 2) * w2 + (x / 2)) + c] = (uint8_t)v;
            }
        }
}

/* Camera::init, src/hpmvs/Camera.cpp:34-81 */
static void camera_init(orc_camera *cam, double f, const double q[4], const double c[3], int width,
                        int height, int max_level) {
    cam->n_levels = max_level + 1;
    float K[9] = {(float)f, 0, (float)(width / 2.0), 0, (float)f, (float)(height / 2.0), 0, 0, 1};
    memcpy(cam->K0, K, sizeof(K));
    /* Eigen::Quaterniond::matrix() in double, then cast<float>() (Camera.cpp:43-50) */
    const double qw = q[0], qx = q[1], qy = q[2], qz = q[3];
    const double tx = 2.0 * qx, ty = 2.0 * qy, tz = 2.0 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
    const double txx = tx * qx, txy = ty * qx, txz = tz * qx;
    const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    float R[9];
    R[0] = (float)(1.0 - (tyy + tzz)); R[1] = (float)(txy - twz); R[2] = (float)(txz + twy);
    R[3] = (float)(txy + twz); R[4] = (float)(1.0 - (txx + tzz)); R[5] = (float)(tyz - twx);
    R[6] = (float)(txz - twy); R[7] = (float)(tyz + twx); R[8] = (float)(1.0 - (txx + tyy));
    const float cf[3] = {(float)c[0], (float)c[1], (float)c[2]};
    float Rt[12];
    for (int r = 0; r < 3; r++) {
        Rt[r * 4 + 0] = R[r * 3 + 0]; Rt[r * 4 + 1] = R[r * 3 + 1]; Rt[r * 4 + 2] = R[r * 3 + 2];
        Rt[r * 4 + 3] = -((R[r * 3 + 0] * cf[0] + R[r * 3 + 1] * cf[1]) + R[r * 3 + 2] * cf[2]);
    }
    /* projection_[0] = kMat_[0] * [R | -R c] (Camera.cpp:52) */
    for (int r = 0; r < 3; r++)
        for (int col = 0; col < 4; col++)
            cam->P[0][r * 4 + col] =
                (K[r * 3 + 0] * Rt[0 * 4 + col] + K[r * 3 + 1] * Rt[1 * 4 + col]) + K[r * 3 + 2] * Rt[2 * 4 + col];
    /* lower levels: rows 0,1 halved (Camera.cpp:55-63) */
    for (int l = 1; l < cam->n_levels; l++) {
        memcpy(cam->P[l], cam->P[l - 1], sizeof(float) * 12);
        for (int k = 0; k < 8; k++) cam->P[l][k] = (float)(cam->P[l][k] / 2.0);
    }
    cam->center[0] = cf[0]; cam->center[1] = cf[1]; cam->center[2] = cf[2]; cam->center[3] = 1.0f;
    /* axes (Camera.cpp:65-76) */
    const float *r2 = &cam->P[0][8];
    const float n = sqrtf(dot3(r2, r2));
    for (int k = 0; k < 4; k++) cam->oaxis[k] = r2[k] / n;
    for (int k = 0; k < 3; k++) cam->zaxis[k] = cam->oaxis[k];
    float xa[3] = {cam->P[0][0], cam->P[0][1], cam->P[0][2]};
    float ya[3];
    cross3(cam->zaxis, xa, ya);
    normalized3(ya, cam->yaxis);
    cross3(cam->yaxis, cam->zaxis, xa);
    normalized3(xa, cam->xaxis);
}

void orc_scene_set_view(orc_scene *s, int i, int w, int h, const uint8_t *rgb_l0, double f,
                        const double q_wxyz[4], const double c[3], int max_level) {
    /* Image::init/load: maxLevel_ = max(1, maxLevel); maxLevel_+1 levels (Image.cpp:33-57) */
    const int img_levels = (max_level > 1 ? max_level : 1) + 1;
    orc_image *im = &s->imgs[i];
    for (int l = 0; l < ORC_MAX_LEVELS; l++) { free(im->pix[l]); im->pix[l] = NULL; }
    im->n_levels = img_levels;
    im->w[0] = w; im->h[0] = h;
    im->pix[0] = (uint8_t *)malloc((size_t)w * h * 3 + 16);
    memcpy(im->pix[0], rgb_l0, (size_t)w * h * 3);
    for (int l = 1; l < img_levels; l++) {
        im->w[l] = im->w[l - 1] / 2; im->h[l] = im->h[l - 1] / 2;
        im->pix[l] = (uint8_t *)calloc((size_t)im->w[l] * im->h[l] * 3 + 16, 1);
        orc_half_resize(im->pix[l - 1], im->w[l - 1], im->h[l - 1], im->pix[l]);
    }
    camera_init(&s->cams[i], f, q_wxyz, c, w, h, max_level);
}

void orc_scene_set_covis(orc_scene *s, int i, const int *ids, int n) {
    /* lists must be set in view order 0..n_views-1 */
    const int off = s->covis_off[i];
    s->covis = (int *)realloc(s->covis, sizeof(int) * (size_t)(off + n + 1));
    memcpy(s->covis + off, ids, sizeof(int) * (size_t)n);
    for (int k = i + 1; k <= s->n_views; k++) s->covis_off[k] = off + n;
}

const orc_camera *orc_scene_camera(const orc_scene *s, int i) { return &s->cams[i]; }
const uint8_t *orc_scene_level(const orc_scene *s, int i, int level, int *w, int *h) {
    *w = s->imgs[i].w[level]; *h = s->imgs[i].h[level];
    return s->imgs[i].pix[level];
}

/* ------------------------------------------------------------------------- */
/* L1 primitives                                                              */

/* Camera::project, include/hpmvs/Camera.h:45-62 */
static void cam_project(const orc_camera *cam, const float *X, int level, float *out) {
    if (level >= cam->n_levels) { fprintf(stderr, "Illegal level access in Camera"); exit(1); }
    const float *P = cam->P[level];
    float r[3];
    for (int k = 0; k < 3; k++)
        r[k] = sum4(P[k * 4 + 0] * X[0], P[k * 4 + 1] * X[1], P[k * 4 + 2] * X[2], P[k * 4 + 3] * X[3]);
    if (r[2] <= 0.0f) {
        out[0] = -65535.0f; out[1] = -65535.0f; out[2] = -1.0f;
    } else {
        const float lo = (float)(INT_MIN + 3.0f), hi = (float)(INT_MAX - 3.0f);
        { const float z = r[2], iz = 1.0f / z; r[0] = vdiv(r[0], z, iz); r[1] = vdiv(r[1], z, iz); r[2] = vdiv(r[2], z, iz); }  /* Eigen: result /= result(2) */
        out[0] = fmaxf(lo, fminf(hi, r[0]));
        out[1] = fmaxf(lo, fminf(hi, r[1]));
        out[2] = r[2];
    }
}

/* Camera::getScale, src/hpmvs/Camera.cpp:83-90 */
static float cam_get_scale(const orc_camera *cam, const float *X, int level) {
    float d[4]; sub4(X, cam->center, d);
    const float fz = norm4(d);
    const float ftmp = cam->K0[0] + cam->K0[4];
    if (ftmp == 0.0) return 1.0f;
    return (float)(2.0 * fz * (0x0001 << level) / ftmp);
}
/* Camera::getLevel, src/hpmvs/Camera.cpp:92-95 */
static float cam_get_level(const orc_camera *cam, const float *X, float scale) {
    float d[4]; sub4(X, cam->center, d);
    const float fz = norm4(d);
    return (float)log2(scale * (float)(cam->K0[0] + cam->K0[4]) / (2.0 * fz));
}
/* Camera::getLeveli, src/hpmvs/Camera.cpp:97-99 */
static int cam_get_leveli(const orc_camera *cam, const float *X, float scale, int max_level) {
    const int l = (int)roundf(cam_get_level(cam, X, scale));
    const int a = l < max_level ? l : max_level;
    return a > 0 ? a : 0;
}

/* Image::getColor, include/hpmvs/Image.h:89-115 */
static void img_get_color(const orc_image *im, float x, float y, int level, float *rgb) {
    const int W = im->w[level];
    const int lx = (int)x, ly = (int)y;
    const long index = 3 * ((long)ly * W + lx);
    const float dx1 = x - lx, dx0 = 1.0f - dx1;
    const float dy1 = y - ly, dy0 = 1.0f - dy1;
    const float f00 = dx0 * dy0, f01 = dx0 * dy1, f10 = dx1 * dy0, f11 = dx1 * dy1;
    const uint8_t *p0 = im->pix[level] + index;
    const uint8_t *p1 = p0 + 3 * (long)W;
    float r = 0.0f, g = 0.0f, b = 0.0f;
    r += p0[0] * f00 + p1[0] * f01;
    g += p0[1] * f00 + p1[1] * f01;
    b += p0[2] * f00 + p1[2] * f01;
    r += p0[3] * f10 + p1[3] * f11;
    g += p0[4] * f10 + p1[4] * f11;
    b += p0[5] * f10 + p1[5] * f11;
    rgb[0] = r; rgb[1] = g; rgb[2] = b;
}

/* Patch2d::normalize, include/hpmvs/Patch2d.hpp:46-84 */
static void tex_normalize(float *d) {
    float ave[3] = {0, 0, 0};
    for (int i = 0; i < 49; i++) { ave[0] += d[3 * i]; ave[1] += d[3 * i + 1]; ave[2] += d[3 * i + 2]; }
    { const float in = 1.0f / 49; ave[0] = vdiv(ave[0], 49, in); ave[1] = vdiv(ave[1], 49, in); ave[2] = vdiv(ave[2], 49, in); }  /* Eigen: ave /= size3 */
    float ave2 = 0.0f;
    for (int i = 0; i < 49; i++) {
        const float f0 = ave[0] - d[3 * i], f1 = ave[1] - d[3 * i + 1], f2 = ave[2] - d[3 * i + 2];
        ave2 += f0 * f0 + f1 * f1 + f2 * f2;
    }
    ave2 = sqrtf(ave2 / 147);
    if (ave2 == 0.0f) ave2 = 1.0f;
    for (int i = 0; i < 49; i++)
        for (int c = 0; c < 3; c++) { d[3 * i + c] -= ave[c]; d[3 * i + c] /= ave2; }
}
/* Patch2d::dot, include/hpmvs/Patch2d.hpp:37-44 */
static float tex_dot(const float *a, const float *b) {
    float ans = 0.0f;
    for (int i = 0; i < 147; i++) ans += a[i] * b[i];
    return ans / 147;
}

/* PatchOptimizer.h:92-98 */
static inline float robustincc(const float rhs) { return rhs / (1 + 3 * rhs); }

/* ------------------------------------------------------------------------- */
/* L2: the PatchOptimizer state (include/hpmvs/PatchOptimizer.h:103-131)      */

typedef struct {
    const orc_scene *s;
    const orc_options *o;
    float pCenter[4], pNormal[4], pScale;
    float pX[4], pY[4], pZ[4];
    int nimg;
    int img[ORC_MAX_IMAGES];
    float refCenter[4], refRay[4];
    float imgX0[3], imgY0[3], imgZ0[3]; /* only index 0 of imgX_/imgY_/imgZ_ is ever read */
    float depthScale, angleScale;
    float refTex[147], comTex[147];
    long ngrabs, ngrabs_all;
    int overflow; /* attached-image list hit ORC_MAX_IMAGES */
    int last_level;
    float last_cdxdy[6];
} popt;

/* PatchOptimizer::calculatePatchAxis, src/hpmvs/PatchOptimizer.cpp:532-548 */
static void calculate_patch_axis(popt *po, int refCam, const float *n, float scale) {
    const orc_camera *rc = &po->s->cams[refCam];
    float x[3], y[3], z[3], t[3];
    normalized3(n, z);
    cross3(z, rc->xaxis, t); normalized3(t, y);
    cross3(y, z, t); normalized3(t, x);
    for (int k = 0; k < 3; k++) { x[k] *= scale; y[k] *= scale; }
    float yn[3], yan[3];
    normalized3(y, yn); normalized3(rc->yaxis, yan);
    const float fs = dot3(yn, yan);
    for (int k = 0; k < 3; k++) y[k] = y[k] * fs;
    for (int k = 0; k < 3; k++) { po->pX[k] = x[k]; po->pY[k] = y[k]; po->pZ[k] = z[k]; }
    po->pX[3] = po->pY[3] = po->pZ[3] = 0.0f;
}

/* PatchOptimizer::sampleTexture, src/hpmvs/PatchOptimizer.cpp:476-529 */
static int sample_texture(popt *po, const float *pCenter, float pScale, const float *pxaxis,
                          const float *pyaxis, const float *pzaxis, int camIdx, float *tex) {
    const orc_image *image = &po->s->imgs[camIdx];
    const orc_camera *camera = &po->s->cams[camIdx];
    po->ngrabs_all++;
    float d[4], dn[4];
    sub4(camera->center, pCenter, d); normalized4(d, dn);
    if (dot4(dn, pzaxis) < cos(po->o->MAX_ANGLE)) return 0;

    const int lvl = cam_get_leveli(camera, pCenter, pScale, po->o->MAXLEVEL - 1);
    float pc[3], px[3], py[3], tmp[4];
    cam_project(camera, pCenter, lvl, pc);
    add4(pCenter, pxaxis, tmp); cam_project(camera, tmp, lvl, px);
    add4(pCenter, pyaxis, tmp); cam_project(camera, tmp, lvl, py);
    const float center[2] = {pc[0], pc[1]};
    const float dx[2] = {px[0] - center[0], px[1] - center[1]};
    const float dy[2] = {py[0] - center[0], py[1] - center[1]};

    const float halfSize = 7 / 2.0f;
    float tl[2], tr[2], bl[2], br[2], mn[2], mx[2];
    for (int k = 0; k < 2; k++) {
        tl[k] = (center[k] - halfSize * dx[k]) - halfSize * dy[k];
        tr[k] = (center[k] + halfSize * dx[k]) - halfSize * dy[k];
        bl[k] = (center[k] - halfSize * dx[k]) + halfSize * dy[k];
        br[k] = (center[k] + halfSize * dx[k]) + halfSize * dy[k];
        mn[k] = fminf(fminf(fminf(tl[k], tr[k]), bl[k]), br[k]);
        mx[k] = fmaxf(fmaxf(fmaxf(tl[k], tr[k]), bl[k]), br[k]);
    }
    const int m = 3;
    if (mn[0] < m || mn[1] < m || mx[0] >= image->w[lvl] - m || mx[1] >= image->h[lvl] - m) return 0;

    po->ngrabs++;
    po->last_level = lvl;
    po->last_cdxdy[0] = center[0]; po->last_cdxdy[1] = center[1];
    po->last_cdxdy[2] = dx[0]; po->last_cdxdy[3] = dx[1];
    po->last_cdxdy[4] = dy[0]; po->last_cdxdy[5] = dy[1];

    float *target = tex;
    float l[2] = {tl[0], tl[1]};
    for (int yy = 0; yy < 7; yy++) {
        float c[2] = {l[0], l[1]};
        l[0] += dy[0]; l[1] += dy[1];
        for (int xx = 0; xx < 7; xx++) {
            float color[3];
            img_get_color(image, c[0], c[1], lvl, color);
            *(target++) = color[0]; *(target++) = color[1]; *(target++) = color[2];
            c[0] += dx[0]; c[1] += dx[1];
        }
    }
    tex_normalize(tex);
    return 1;
}

/* PatchOptimizer::setINCCs, src/hpmvs/PatchOptimizer.cpp:448-474 */
static void set_inccs(popt *po, float *inccs, const int *indexes, int n, int refIdx, int robust) {
    calculate_patch_axis(po, indexes[refIdx], po->pNormal, po->pScale);
    if (!sample_texture(po, po->pCenter, po->pScale, po->pX, po->pY, po->pNormal, indexes[refIdx], po->refTex)) {
        for (int i = 0; i < n; i++) inccs[i] = 2.0f;
        return;
    }
    for (int ii = 0; ii < n; ii++) {
        if (ii == refIdx) inccs[ii] = 0.0f;
        else if (!sample_texture(po, po->pCenter, po->pScale, po->pX, po->pY, po->pNormal, indexes[ii], po->comTex))
            inccs[ii] = 2.0f;
        else if (robust) inccs[ii] = robustincc(1.0f - tex_dot(po->refTex, po->comTex));
        else inccs[ii] = 1.0f - tex_dot(po->refTex, po->comTex);
    }
}

/* PatchOptimizer::addImages, src/hpmvs/PatchOptimizer.cpp:225-258 */
static int add_images(popt *po) {
    if (po->nimg <= 0) return 0;
    const orc_scene *s = po->s;
    const int refImg = po->img[0];
    const int n_existing = po->nimg;
    for (int ci = s->covis_off[refImg]; ci < s->covis_off[refImg + 1]; ci++) {
        const int cv = s->covis[ci];
        int found = 0;
        for (int k = 0; k < n_existing; k++) if (po->img[k] == cv) { found = 1; break; }
        if (found) continue;
        const orc_camera *cam = &s->cams[cv];
        float d[4], dn[4];
        sub4(cam->center, po->pCenter, d); normalized4(d, dn);
        if (dot4(dn, po->pNormal) < cosf(po->o->MAX_ANGLE)) continue;
        const int imgLevel = (int)roundf(cam_get_level(cam, po->pCenter, po->pScale));
        if (imgLevel < po->o->MINLEVEL || imgLevel >= po->o->MAXLEVEL - 2) continue;
        float pc[3];
        cam_project(cam, po->pCenter, imgLevel, pc);
        if (pc[0] < 0.0f || s->imgs[cv].w[imgLevel] - 1 <= pc[0] || pc[1] < 0.0f ||
            s->imgs[cv].h[imgLevel] - 1 <= pc[1])
            continue;
        /* the list is a fixed ORC_MAX_IMAGES array here (reference: unbounded vector<int>): an image that does
         * not fit fails the patch with stage 100, exactly as the device path does */
        if (po->nimg >= ORC_MAX_IMAGES) { po->overflow = 1; return 0; }
        po->img[po->nimg++] = cv;
    }
    return po->nimg >= po->o->MIN_IMAGES_PER_PATCH;
}

/* PatchOptimizer::filterImagesNCC, src/hpmvs/PatchOptimizer.cpp:138-152 */
static int filter_images_ncc(popt *po, float threshold) {
    float inccs[ORC_MAX_IMAGES];
    set_inccs(po, inccs, po->img, po->nimg, 0, 0);
    int n = 1;
    for (int i = 1; i < po->nimg; i++)
        if (inccs[i] < 1.0f - threshold) po->img[n++] = po->img[i];
    po->nimg = n;
    return po->nimg >= po->o->MIN_IMAGES_PER_PATCH;
}

/* PatchOptimizer::sortImages + getAngleWeightedScales, src/hpmvs/PatchOptimizer.cpp:183-223, 260-284 */
static int sort_images(popt *po) {
    const float threshold = (float)(1.0f - cos(10.0 * M_PI / 180.0));
    int indexes[ORC_MAX_IMAGES], n = 0;
    float wScales[ORC_MAX_IMAGES];
    float rays[ORC_MAX_IMAGES][4];
    if (po->nimg > 0) {
        const orc_camera *c0 = &po->s->cams[po->img[0]];
        int refLevel = (int)roundf(cam_get_level(c0, po->pCenter, po->pScale));
        if (refLevel > po->o->MAXLEVEL - 1) refLevel = po->o->MAXLEVEL - 1;
        if (refLevel < 0) refLevel = 0;
        float nn[4]; normalized4(po->pNormal, nn);
        for (int k = 0; k < po->nimg; k++) {
            const orc_camera *cam = &po->s->cams[po->img[k]];
            float d[4], ray[4];
            sub4(cam->center, po->pCenter, d); normalized4(d, ray);
            const float cosa = dot4(ray, nn);
            if (cosa > 0) {
                indexes[n] = po->img[k];
                memcpy(rays[n], ray, sizeof(ray));
                wScales[n] = cam_get_scale(cam, po->pCenter, refLevel) / cosa;
                n++;
            }
        }
    }
    po->nimg = 0;
    if (n < 2) return 0;
    wScales[0] = 0.0f;
    while (n > 0) {
        int index = 0;
        for (int j = 1; j < n; j++) if (wScales[j] < wScales[index]) index = j;
        po->img[po->nimg++] = indexes[index];
        float ri[4]; memcpy(ri, rays[index], sizeof(ri));
        int m = 0;
        for (int j = 0; j < n; j++) {
            if (j == index) continue;
            const float ftmp = fminf(threshold, fmaxf(threshold / 2.0f, 1.0f - dot4(ri, rays[j])));
            const float ws = wScales[j] * (threshold / ftmp);
            indexes[m] = indexes[j];
            if (m != j) memcpy(rays[m], rays[j], sizeof(ri));
            wScales[m] = ws;
            m++;
        }
        n = m;
    }
    return po->nimg >= po->o->MIN_IMAGES_PER_PATCH;
}

/* PatchOptimizer::assureImageAngles, src/hpmvs/PatchOptimizer.cpp:105-123 */
static int assure_image_angles(popt *po) {
    float rays[ORC_MAX_IMAGES][4];
    for (int k = 0; k < po->nimg; k++) {
        float d[4]; sub4(po->s->cams[po->img[k]].center, po->pCenter, d); normalized4(d, rays[k]);
    }
    for (int ii = 0; ii < po->nimg - 1; ii++)
        for (int jj = ii + 1; jj < po->nimg; jj++) {
            const float a = acosf(dot4(rays[ii], rays[jj]));
            if (a < po->o->MAX_ANGLE && a > po->o->MIN_ANGLE) return 1;
        }
    return 0;
}

/* PatchOptimizer::filterImagesByAngle, src/hpmvs/PatchOptimizer.cpp:125-136 */
static int filter_images_by_angle(popt *po) {
    int n = 0;
    for (int k = 0; k < po->nimg; k++) {
        float d[4], dn[4];
        sub4(po->s->cams[po->img[k]].center, po->pCenter, d); normalized4(d, dn);
        if (dot4(dn, po->pNormal) > cosf(po->o->MAX_ANGLE)) po->img[n++] = po->img[k];
    }
    po->nimg = n;
    return po->nimg >= po->o->MIN_IMAGES_PER_PATCH;
}

/* PatchOptimizer::setRefImage, src/hpmvs/PatchOptimizer.cpp:154-181 */
static void set_ref_image(popt *po) {
    if (po->nimg <= 1) return;
    float incc[ORC_MAX_IMAGES];
    int refindex = -1;
    float refncc = 3.402823466e+38f;
    for (int ii = 0; ii < po->nimg; ii++) {
        set_inccs(po, incc, po->img, po->nimg, ii, 1);
        float sum = 0.0f;
        for (int k = 0; k < po->nimg; k++) sum += incc[k];
        if (sum < refncc) { refncc = sum; refindex = ii; }
    }
    if (refindex < 0) return; /* reference would index out of bounds; cannot happen (sums <= 2V) */
    const int refIndex = po->img[refindex];
    for (int i = 0; i < po->nimg; i++)
        if (po->img[i] == refIndex) {
            const int itmp = po->img[0];
            po->img[0] = refIndex;
            po->img[i] = itmp;
            break;
        }
}

/* PatchOptimizer::setOptimizationFields, src/hpmvs/PatchOptimizer.cpp:384-399 */
static void set_optimization_fields(popt *po) {
    const orc_camera *c0 = &po->s->cams[po->img[0]];
    normalized3(c0->xaxis, po->imgX0);
    normalized3(c0->yaxis, po->imgY0);
    normalized3(c0->zaxis, po->imgZ0);
    memcpy(po->refCenter, po->pCenter, sizeof(float) * 4);
    float d[4]; sub4(po->refCenter, c0->center, d); normalized4(d, po->refRay);
    po->depthScale = 1.0f;
    po->angleScale = (float)(M_PI / 48.0f);
}

/* PatchOptimizer::setCenterNorm, src/hpmvs/PatchOptimizer.cpp:401-414 */
static void set_center_norm(popt *po, const double *x) {
    const float x0 = (float)x[0];
    for (int k = 0; k < 4; k++) po->pCenter[k] = po->refCenter[k] + (x0 * po->refRay[k]) * po->depthScale;
    const float angle1 = (float)(x[1] * po->angleScale);
    const float angle2 = (float)(x[2] * po->angleScale);
    double s1, c1, s2, c2;  /* (float arguments promoted to double: the unqualified sin / cos of the reference, see the header) */
    path_sincos((double)angle1, &s1, &c1);
    path_sincos((double)angle2, &s2, &c2);
    const float fx = (float)(s1 * c2);
    const float fy = (float)s2;
    const float fz = (float)(-c1 * c2);
    for (int k = 0; k < 3; k++)
        po->pNormal[k] = (po->imgX0[k] * fx + po->imgY0[k] * fy) + po->imgZ0[k] * fz;
    po->pNormal[3] = 0.0f;
}

/* PatchOptimizer::parametersFromCenterNorm, src/hpmvs/PatchOptimizer.cpp:416-446 */
static void parameters_from_center_norm(popt *po, const float *c, const float *n, const double *lb,
                                        const double *ub, double *x) {
    float d[4]; sub4(c, po->refCenter, d);
    x[0] = dot4(d, po->refRay) / po->depthScale;
    const float fx = dot3(po->imgX0, n), fy = dot3(po->imgY0, n), fz = dot3(po->imgZ0, n);
    x[2] = asinf(fy);
    /* std::max(-1.0, std::min(1.0, x[2])) with the std:: NaN semantics (first argument wins) */
    double t = (x[2] < 1.0) ? x[2] : 1.0;
    t = (-1.0 < t) ? t : -1.0;
    const float cosb = (float)cos(t);
    if (cosb == 0.0) x[1] = 0.0;
    else {
        const double sina = fx / cosb;
        const double cosa = -fz / cosb;
        double u = (-1.0 < cosa) ? cosa : -1.0;
        u = (u < 1.0) ? u : 1.0;
        x[1] = path_acos(u);
        if (sina < 0.0) x[1] = -x[1];
    }
    x[1] /= po->angleScale;
    x[2] /= po->angleScale;
    for (int k = 0; k < 3; k++) {
        double v = (lb[k] < x[k]) ? x[k] : lb[k]; /* std::max(lb, x) */
        x[k] = (v < ub[k]) ? v : ub[k];           /* std::min(ub, .) */
    }
}

/* PatchOptimizer::objective_fn, src/hpmvs/PatchOptimizer.cpp:286-311 */
static double objective_fn(popt *po) {
    calculate_patch_axis(po, po->img[0], po->pNormal, po->pScale);
    if (!sample_texture(po, po->pCenter, po->pScale, po->pX, po->pY, po->pZ, po->img[0], po->refTex)) return 2.0;
    double val = 0.0;
    int nImgs = 0;
    for (int ii = 1; ii < po->nimg; ii++) {
        if (!sample_texture(po, po->pCenter, po->pScale, po->pX, po->pY, po->pZ, po->img[ii], po->comTex)) continue;
        val += robustincc((float)(1.0 - tex_dot(po->refTex, po->comTex)));
        nImgs++;
    }
    if (nImgs < po->o->MIN_IMAGES_PER_PATCH - 1) return 2.0;
    return val / nImgs;
}

/* PatchOptimizer::static_objective_fn, src/hpmvs/PatchOptimizer.cpp:313-320 */
static double static_objective_fn(unsigned n, const double *x, double *grad, void *data) {
    (void)n; (void)grad;
    popt *po = (popt *)data;
    set_center_norm(po, x);
    return objective_fn(po);
}

/* ------------------------------------------------------------------------- */
/* L1': NLopt front end (N1-N3) around BOBYQA                                  */

typedef int (*ref_bobyqa_fn)(int n, int npt, double *x, const double *lb, const double *ub, const double *dx,
                             double xtol_rel, int maxeval, double *minf, int *nevals, orc_obj_fn f, void *data);
/* tests/native/bobyqa_port.cpp -> tests/native/libbq_port.so: a host build of the PRODUCT's BOBYQA state machine.  It is NOT part
 * of this library (an oracle holds no product code): it is loaded at run time, and only when a caller names ORC_OPT_PORT. */
static ref_bobyqa_fn g_port_bobyqa = NULL;
static int g_port_tried = 0;
static ref_bobyqa_fn load_port_bobyqa(void);
static ref_bobyqa_fn g_ref_bobyqa = NULL;
static int g_ref_tried = 0;

/* (first use may come from many OpenMP threads at once: the flag goes up only after the pointer is in place) */
static ref_bobyqa_fn load_ref_bobyqa(void) {
    if (__atomic_load_n(&g_ref_tried, __ATOMIC_ACQUIRE) == 2) return g_ref_bobyqa;
    int expected = 0;
    if (!__atomic_compare_exchange_n(&g_ref_tried, &expected, 1, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
        while (__atomic_load_n(&g_ref_tried, __ATOMIC_ACQUIRE) != 2) { }
        return g_ref_bobyqa;
    }
    Dl_info info;
    char path[4096];
    if (dladdr((void *)&load_ref_bobyqa, &info) && info.dli_fname) {
        snprintf(path, sizeof(path), "%s", info.dli_fname);
        char *slash = strrchr(path, '/');
        if (slash) *slash = 0; else snprintf(path, sizeof(path), ".");
        strncat(path, "/_ref/libref_bobyqa.so", sizeof(path) - strlen(path) - 1);
        void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (h) g_ref_bobyqa = (ref_bobyqa_fn)dlsym(h, "ref_bobyqa_run");
    }
    __atomic_store_n(&g_ref_tried, 2, __ATOMIC_RELEASE);
    return g_ref_bobyqa;
}

static ref_bobyqa_fn load_port_bobyqa(void) {
    if (__atomic_load_n(&g_port_tried, __ATOMIC_ACQUIRE) == 2) return g_port_bobyqa;
    int expected = 0;
    if (!__atomic_compare_exchange_n(&g_port_tried, &expected, 1, 0, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
        while (__atomic_load_n(&g_port_tried, __ATOMIC_ACQUIRE) != 2) { }
        return g_port_bobyqa;
    }
    Dl_info info;
    char path[4096];
    if (dladdr((void *)&load_port_bobyqa, &info) && info.dli_fname) {
        snprintf(path, sizeof(path), "%s", info.dli_fname);
        char *slash = strrchr(path, '/');
        if (slash) *slash = 0; else snprintf(path, sizeof(path), ".");
        strncat(path, "/../tests/native/libbq_port.so", sizeof(path) - strlen(path) - 1);
        void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (h) g_port_bobyqa = (ref_bobyqa_fn)dlsym(h, "orc_port_bobyqa");
    }
    __atomic_store_n(&g_port_tried, 2, __ATOMIC_RELEASE);
    return g_port_bobyqa;
}

int orc_optimizer_available(int which) {
    if (which == ORC_OPT_REF) return load_ref_bobyqa() ? 0 : -1;
    return load_port_bobyqa() ? 0 : -1;
}

static int is_inf(double x) { return fabs(x) >= HUGE_VAL * 0.99 || isinf(x); } /* api/general.c:30-36 */

/* nlopt_optimize_ for LN_BOBYQA: thirdLibs/nlopt-2.4.2/api/optimize.c:368-416, 669-681 with the
 * default initial step of api/options.c:686-728.  Returns the nlopt_result code. */
int orc_bobyqa_minimize(int which, int n, orc_obj_fn f, void *data, double *x, const double *lb,
                        const double *ub, double xtol_rel, int maxeval, double *minf, int *nevals) {
    *minf = HUGE_VAL;
    *nevals = 0;
    for (int i = 0; i < n; i++)
        if (lb[i] > ub[i] || x[i] < lb[i] || x[i] > ub[i]) return -2; /* NLOPT_INVALID_ARGS */
    double dx[16];
    for (int i = 0; i < n; i++) {
        double step = HUGE_VAL;
        if (!is_inf(ub[i]) && !is_inf(lb[i]) && (ub[i] - lb[i]) * 0.25 < step && ub[i] > lb[i])
            step = (ub[i] - lb[i]) * 0.25;
        if (!is_inf(ub[i]) && ub[i] - x[i] < step && ub[i] > x[i]) step = (ub[i] - x[i]) * 0.75;
        if (!is_inf(lb[i]) && x[i] - lb[i] < step && x[i] > lb[i]) step = (x[i] - lb[i]) * 0.75;
        if (is_inf(step)) {
            if (!is_inf(ub[i]) && fabs(ub[i] - x[i]) < fabs(step)) step = (ub[i] - x[i]) * 1.1;
            if (!is_inf(lb[i]) && fabs(x[i] - lb[i]) < fabs(step)) step = (x[i] - lb[i]) * 1.1;
        }
        if (is_inf(step) || step == 0) step = x[i];
        if (is_inf(step) || step == 0) step = 1;
        dx[i] = step;
    }
    if (which == ORC_OPT_REF) {
        ref_bobyqa_fn fn = load_ref_bobyqa();
        if (!fn) { fprintf(stderr, "oracle: oracle/_ref/libref_bobyqa.so missing\n"); return -1; }
        return fn(n, 2 * n + 1, x, lb, ub, dx, xtol_rel, maxeval, minf, nevals, f, data);
    }
    ref_bobyqa_fn pf = load_port_bobyqa();
    if (!pf) { fprintf(stderr, "oracle: tests/native/libbq_port.so missing (make -C tests/native)\n"); return -1; }
    return pf(n, 2 * n + 1, x, lb, ub, dx, xtol_rel, maxeval, minf, nevals, f, data);
}

/* PatchOptimizer::optimizePatch, src/hpmvs/PatchOptimizer.cpp:322-382 */
static int optimize_patch(popt *po, int which, orc_patch *diag) {
    if (po->nimg < po->o->MIN_IMAGES_PER_PATCH) return 0;
    const double min_angle = -23.99999, max_angle = 23.99999;
    const double lb[3] = {-HUGE_VAL, min_angle, min_angle};
    const double ub[3] = {HUGE_VAL, max_angle, max_angle};
    double x[3] = {0, 0, 0};
    set_optimization_fields(po);
    parameters_from_center_norm(po, po->refCenter, po->pNormal, lb, ub, x);
    double minf;
    int nevals = 0;
    const int result = orc_bobyqa_minimize(which, 3, static_objective_fn, po, x, lb, ub, 1.e-7, 1000, &minf, &nevals);
    diag->nlopt_result = result;
    diag->nevals = nevals;
    diag->fmin = minf;
    diag->x[0] = x[0]; diag->x[1] = x[1]; diag->x[2] = x[2];
    /* SUCCESS, STOPVAL_REACHED, FTOL_REACHED, XTOL_REACHED; negative codes throw in nlopt.hpp:138-147 */
    const int success = (result == 1 || result == 2 || result == 3 || result == 4);
    if (!success) return 0;
    set_center_norm(po, x);
    return 1;
}

/* Scene::getColor(const Patch3d&), src/hpmvs/Scene.cpp:300-327 */
static void scene_get_color(const orc_scene *s, const float *center, float scale, const int *img, int nimg,
                            float *out) {
    float colors[ORC_MAX_IMAGES][3];
    float norms[ORC_MAX_IMAGES];
    for (int k = 0; k < nimg; k++) {
        const orc_camera *cam = &s->cams[img[k]];
        const int lvl = cam_get_leveli(cam, center, scale, cam->n_levels - 1);
        float c[3];
        cam_project(cam, center, lvl, c);
        img_get_color(&s->imgs[img[k]], c[0], c[1], lvl, colors[k]);
        norms[k] = sqrtf(dot3(colors[k], colors[k]));
    }
    /* std::sort by norm: insertion sort (stable; ties are measure-zero) */
    int order[ORC_MAX_IMAGES];
    for (int k = 0; k < nimg; k++) order[k] = k;
    for (int i = 1; i < nimg; i++) {
        const int v = order[i];
        int j = i - 1;
        while (j >= 0 && norms[order[j]] > norms[v]) { order[j + 1] = order[j]; j--; }
        order[j + 1] = v;
    }
    const int med = order[nimg / 2];
    const int pick = (norms[med] > 250.0) ? order[0] : med;
    out[0] = colors[pick][0]; out[1] = colors[pick][1]; out[2] = colors[pick][2];
}

static void popt_begin(popt *po, const orc_scene *s, const orc_options *o, const orc_patch *p) {
    memset(po, 0, sizeof(*po));
    po->s = s; po->o = o;
    memcpy(po->pCenter, p->center, sizeof(float) * 4);
    memcpy(po->pNormal, p->normal, sizeof(float) * 4);
    po->pScale = p->scale;
    po->nimg = p->n_images;
    memcpy(po->img, p->images, sizeof(int) * (size_t)p->n_images);
}

/* PatchOptimizer::optimize + runOptimization, src/hpmvs/PatchOptimizer.cpp:48-103 */
int orc_optimize(const orc_scene *s, const orc_options *o, orc_patch *p, int which) {
    popt po;
    popt_begin(&po, s, o, p);
    p->nlopt_result = 0; p->nevals = 0; p->fmin = 0; p->x[0] = p->x[1] = p->x[2] = 0;
    int stage = 0;
    for (int k = 0; k < p->n_images; k++)
        if (p->images[k] < 0 || p->images[k] >= s->n_views) { p->stage = 101; p->ngrabs = 0; p->ngrabs_all = 0; return 0; } /* as the device path */
    do {
        stage = 1; if (!add_images(&po)) break;
        stage = 2; if (!filter_images_ncc(&po, o->NCC_ALPHA_1)) break;
        sort_images(&po); /* return value ignored, PatchOptimizer.cpp:54 */
        stage = 3; if (!assure_image_angles(&po)) break;
        stage = 4; if (!optimize_patch(&po, which, p)) break;
        stage = 5; if (!add_images(&po)) break;
        stage = 6; if (!filter_images_ncc(&po, o->NCC_ALPHA_2)) break;
        stage = 7; if (!filter_images_by_angle(&po)) break;
        stage = 8; if (!assure_image_angles(&po)) break;
        set_ref_image(&po);
        stage = 9; if (!filter_images_ncc(&po, o->NCC_ALPHA_2)) break;
        stage = 0;
    } while (0);
    if (po.overflow) stage = 100;
    p->stage = stage;
    p->ngrabs = po.ngrabs;
    p->ngrabs_all = po.ngrabs_all;
    if (stage != 0) return 0; /* patch untouched on failure (PatchOptimizer.cpp:86-87) */
    memcpy(p->center, po.pCenter, sizeof(float) * 4);
    memcpy(p->normal, po.pNormal, sizeof(float) * 4);
    p->scale = po.pScale;
    p->n_images = po.nimg;
    memcpy(p->images, po.img, sizeof(int) * (size_t)po.nimg);
    p->ncc_const = 1.4f;
    scene_get_color(s, p->center, p->scale, p->images, p->n_images, p->color);
    return 1;
}

void orc_optimize_batch(const orc_scene *s, const orc_options *o, orc_patch *p, int n, int which, int n_threads) {
    /* the reference's data-parallel form: src/hpmvs/Scene.cpp:94-96,114,166-167 */
    (void)n_threads;
#pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads > 0 ? n_threads : 1)
    for (int i = 0; i < n; i++) orc_optimize(s, o, &p[i], which);
}

/* Scene::initPatches, src/hpmvs/Scene.cpp:112-178 */
void orc_init_patches(const orc_scene *s, const orc_options *o, int start_level, int n, const double *xyz,
                      const int *meas_off, const int *meas_img, orc_patch *pp, int which, int n_threads) {
#pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads > 0 ? n_threads : 1)
    for (int ii = 0; ii < n; ii++) {
        orc_patch *p = &pp[ii];
        memset(p, 0, sizeof(*p));
        p->center[0] = (float)xyz[3 * ii]; p->center[1] = (float)xyz[3 * ii + 1]; p->center[2] = (float)xyz[3 * ii + 2];
        p->center[3] = 1.0f;
        const int m0 = meas_off[ii], m1 = meas_off[ii + 1];
        if (m1 - m0 < o->MIN_IMAGES_PER_PATCH) { p->stage = 10; continue; }
        const int margin = 2; /* cSize */
        int nimg = 0;
        for (int m = m0; m < m1; m++) {
            const int idx = meas_img[m];
            if (idx < 0) continue;
            float pr[3];
            cam_project(&s->cams[idx], p->center, start_level, pr);
            if (pr[0] < margin || pr[1] < margin || pr[0] >= s->imgs[idx].w[start_level] - margin ||
                pr[1] >= s->imgs[idx].h[start_level] - margin)
                continue;
            if (nimg < ORC_MAX_IMAGES) p->images[nimg] = idx;
            nimg++;
        }
        if (nimg < 2) { p->stage = 11; continue; }
        if (nimg > ORC_MAX_IMAGES) { p->stage = 100; continue; }
        p->n_images = nimg;
        const orc_camera *c0 = &s->cams[p->images[0]];
        float d[4];
        sub4(c0->center, p->center, d);
        normalized4(d, p->normal); /* ppatch->normal_.normalize() */
        p->normal[3] = 0.0f;
        p->scale = cam_get_scale(c0, p->center, start_level);
        const float before[3] = {p->center[0], p->center[1], p->center[2]};
        if (!orc_optimize(s, o, p, which)) continue;
        const float dd[3] = {p->center[0] - before[0], p->center[1] - before[1], p->center[2] - before[2]};
        if (sqrtf(dot3(dd, dd)) > p->scale * 2) { p->stage = 12; }
    }
}

/* ---- CellProcessor::extend / ::branch candidate loops ------------------------------------ */

/* Cell::contains, src/hpmvs/doctree.cpp:38-42 */
static int cell_contains(const float *cc, float width, const float *p) {
    const float hw = width / 2.0;
    return p[0] > cc[0] - hw && p[1] > cc[1] - hw && p[2] > cc[2] - hw && p[0] <= cc[0] + hw && p[1] <= cc[1] + hw &&
           p[2] <= cc[2] + hw;
}

int orc_expand_directions(int mode, float *dxdy) {
    const int N = mode == 0 ? 6 : 4;
    for (int ii = 0; ii < N; ii++) {
        /* CellProcessor.cpp:107-109 / 233-235: `const float angle = ...; cos(angle)` with `using namespace std`
         * in scope resolves to std::cos(float) */
        const float angle = mode == 0 ? 2.0 * M_PI / N * ii : 2.0 * M_PI / N * ii + M_PI / 4;
        dxdy[2 * ii] = cosf(angle);
        dxdy[2 * ii + 1] = sinf(angle);
    }
    return N;
}

int orc_expand(const orc_scene *s, const orc_options *o, int mode, const orc_patch *parent, const float cell_center[3],
               float cell_width, const unsigned char *skip, orc_patch *out, int which) {
    const int N = mode == 0 ? 6 : 4;
    float dxdy[12];
    orc_expand_directions(mode, dxdy);
    const orc_patch *p = parent;
    for (int ii = 0; ii < N; ii++) {
        orc_patch *q = &out[ii];
        *q = *p; /* *newP = *p */
        q->stage = 0; q->fmin = 0.0; q->nevals = 0; q->nlopt_result = 0; q->ngrabs = 0; q->ngrabs_all = 0;
        q->x[0] = q->x[1] = q->x[2] = 0.0;
        if (p->n_images <= 0) { q->stage = 20; continue; }
        /* CellProcessor.cpp:97-101 / 223-227 */
        const float *imgX = s->cams[p->images[0]].xaxis;
        float t[3], yaxis[3], xaxis[3];
        cross3(p->normal, imgX, t);
        normalized3(t, yaxis);
        cross3(yaxis, p->normal, xaxis);
        const float extend = mode == 0 ? cell_width : (float)(cell_width / 4.0);
        const float dx = dxdy[2 * ii], dy = dxdy[2 * ii + 1];
        for (int k = 0; k < 3; k++) q->center[k] = p->center[k] + (dx * xaxis[k] + dy * yaxis[k]) * extend;
        q->scale = mode == 0 ? cell_width * 0.9 / 2.0 : cell_width * 0.45 / 2.0;
        if (skip && skip[ii]) { q->stage = 20; continue; }
        if (mode == 1 && !cell_contains(cell_center, cell_width, q->center)) { q->stage = 20; continue; } /* :247 */
        if (!orc_optimize(s, o, q, which)) continue; /* stage set by optimize */
        if (mode == 0) {
            /* :131-133 */
            if (!(q->scale * 2.0 < cell_width && q->scale * 2.0 > cell_width / 2.0)) { q->stage = 21; continue; }
            float d[4];
            sub4(q->center, p->center, d);
            if (!(norm4(d) < cell_width * 1.5)) { q->stage = 22; continue; }
        } else if (!cell_contains(cell_center, cell_width, q->center)) { q->stage = 22; continue; } /* :257 */
    }
    return N;
}

void orc_expand_batch(const orc_scene *s, const orc_options *o, int mode, int n, const orc_patch *parents,
                      const float *cell_center, const float *cell_width, const unsigned char *skip, orc_patch *out,
                      int which, int n_threads) {
    const int N = mode == 0 ? 6 : 4;
#pragma omp parallel for schedule(dynamic, 8) num_threads(n_threads > 0 ? n_threads : 1)
    for (int i = 0; i < n; i++)
        orc_expand(s, o, mode, &parents[i], cell_center + 3 * i, cell_width[i], skip ? skip + (size_t)i * N : NULL,
                   out + (size_t)i * N, which);
}

double orc_objective_at(const orc_scene *s, const orc_options *o, const orc_patch *p, const double x[3]) {
    popt po;
    popt_begin(&po, s, o, p);
    set_optimization_fields(&po);
    return static_objective_fn(3, x, NULL, &po);
}

void orc_initial_parameters(const orc_scene *s, const orc_options *o, const orc_patch *p, double x[3]) {
    popt po;
    popt_begin(&po, s, o, p);
    set_optimization_fields(&po);
    const double lb[3] = {-HUGE_VAL, -23.99999, -23.99999}, ub[3] = {HUGE_VAL, 23.99999, 23.99999};
    parameters_from_center_norm(&po, po.refCenter, po.pNormal, lb, ub, x);
}

void orc_inccs(const orc_scene *s, const orc_options *o, const orc_patch *p, int ref_idx, int robust, float *out) {
    popt po;
    popt_begin(&po, s, o, p);
    set_inccs(&po, out, po.img, po.nimg, ref_idx, robust);
}

int orc_sample_texture(const orc_scene *s, const orc_options *o, const orc_patch *p, int ref_slot, int slot,
                       float *tex147, int *level, float *cdxdy6) {
    popt po;
    popt_begin(&po, s, o, p);
    calculate_patch_axis(&po, po.img[ref_slot], po.pNormal, po.pScale);
    const int ok = sample_texture(&po, po.pCenter, po.pScale, po.pX, po.pY, po.pZ, po.img[slot], tex147);
    if (ok) { *level = po.last_level; memcpy(cdxdy6, po.last_cdxdy, sizeof(float) * 6); }
    return ok;
}

/* ------------------------------------------------------------------------- */
/* Depth maps and the acceptance gates that read them (SURVEY 8f-3's second half).
 *
 * Storage: per view and pyramid level an Eigen::MatrixXf(rows, cols) filled with MAX_DEPTH = 1000
 * (src/hpmvs/Scene.cpp:33,74-80), rows = height(level) / DEPTH_SUBSAMPLE, cols = width(level) / DEPTH_SUBSAMPLE
 * with `const double DEPTH_SUBSAMPLE = 2` (include/hpmvs/Scene.h:78): an int divided by a double, truncated.
 * Eigen's default storage is column-major: element (y, x) is data[y + x * rows]; this struct keeps that layout. */
#define ORC_MAX_DEPTH 1000.0f
struct orc_depths_s {
    int n_views;
    int n_levels;
    int *rows, *cols;   /* [view * ORC_MAX_LEVELS + level] */
    float **d;
};
static inline int sub_index(int v) { return (int)((double)v / 2.0); }  /* `int / DEPTH_SUBSAMPLE` -> int: truncation towards zero */

orc_depths *orc_depths_create(const orc_scene *s) {
    orc_depths *D = (orc_depths *)calloc(1, sizeof(orc_depths));
    D->n_views = s->n_views;
    D->rows = (int *)calloc((size_t)s->n_views * ORC_MAX_LEVELS, sizeof(int));
    D->cols = (int *)calloc((size_t)s->n_views * ORC_MAX_LEVELS, sizeof(int));
    D->d = (float **)calloc((size_t)s->n_views * ORC_MAX_LEVELS, sizeof(float *));
    for (int v = 0; v < s->n_views; v++)
        for (int l = 0; l < s->cams[v].n_levels; l++) {
            const int k = v * ORC_MAX_LEVELS + l;
            D->rows[k] = sub_index(s->imgs[v].h[l]);
            D->cols[k] = sub_index(s->imgs[v].w[l]);
            const size_t n = (size_t)D->rows[k] * D->cols[k];
            D->d[k] = (float *)malloc((n ? n : 1) * sizeof(float));
            for (size_t i = 0; i < n; i++) D->d[k][i] = ORC_MAX_DEPTH;
        }
    return D;
}
void orc_depths_destroy(orc_depths *D) {
    if (!D) return;
    for (int k = 0; k < D->n_views * ORC_MAX_LEVELS; k++) free(D->d[k]);
    free(D->d); free(D->rows); free(D->cols); free(D);
}
float *orc_depths_level(orc_depths *D, int view, int level, int *rows, int *cols) {
    const int k = view * ORC_MAX_LEVELS + level;
    if (rows) *rows = D->rows[k];
    if (cols) *cols = D->cols[k];
    return D->d[k];
}
/* Camera::mult, include/hpmvs/Camera.h:76-78 */
static void cam_mult(const orc_camera *cam, const float *X, int level, float *r) {
    const float *P = cam->P[level];
    for (int k = 0; k < 3; k++)
        r[k] = sum4(P[k * 4 + 0] * X[0], P[k * 4 + 1] * X[1], P[k * 4 + 2] * X[2], P[k * 4 + 3] * X[3]);
}
/* Scene::setDepths(patch, subtract), src/hpmvs/Scene.cpp:351-381: subtract = false offers the patch's depth to one cell per
 * attached image (a minimum); subtract = true takes it back -- a cell that still holds exactly this patch's depth becomes
 * MAX_DEPTH again (CellProcessor::branch does that for the patch of a leaf it splits, CellProcessor.cpp:276-279) */
void orc_set_depths_ex(const orc_scene *s, orc_depths *D, const orc_patch *p, int subtract) {
    for (int i = 0; i < p->n_images; i++) {
        const int idx = p->images[i];
        const orc_camera *cam = &s->cams[idx];
        const int level = cam_get_leveli(cam, p->center, p->scale, cam->n_levels - 1);
        float c[3];
        cam_mult(cam, p->center, level, c);
        const int x = sub_index((int)(c[0] / c[2] + 0.5));
        const int y = sub_index((int)(c[1] / c[2] + 0.5));
        const float d = c[2];
        const int k = idx * ORC_MAX_LEVELS + level;
        if (x < 0 || x >= D->cols[k] || y < 0 || y >= D->rows[k]) continue;
        float *cell = &D->d[k][y + (size_t)x * D->rows[k]];
        const float old = *cell;
        if (old == d && subtract) *cell = ORC_MAX_DEPTH;
        else if (!subtract && d < old) *cell = d;
    }
}
void orc_set_depths(const orc_scene *s, orc_depths *D, const orc_patch *p) { orc_set_depths_ex(s, D, p, 0); }
/* Scene::getLevelSupport(patch, minLevel), src/hpmvs/Scene.cpp:334-343: the attached images in which the patch could still be
 * seen on a finer pyramid level */
int orc_level_support(const orc_scene *s, const orc_patch *p, int min_level) {
    int n = 0;
    for (int i = 0; i < p->n_images; i++)
        if (roundf(cam_get_level(&s->cams[p->images[i]], p->center, p->scale)) > (float)min_level) n++;
    return n;
}
/* Scene::getFullDepth, src/hpmvs/Scene.cpp:406-432 */
static float get_full_depth(const orc_scene *s, const orc_depths *D, int img, int xx, int yy) {
    float depth = ORC_MAX_DEPTH;
    int x = sub_index(xx), y = sub_index(yy);
    for (int level = 0; level < s->cams[img].n_levels; level++) {
        const int k = img * ORC_MAX_LEVELS + level;
        if (x < 0 || x >= D->cols[k] || y < 0 || y >= D->rows[k]) return depth;
        const float v = D->d[k][y + (size_t)x * D->rows[k]];
        depth = v < depth ? v : depth;  /* std::min(depth, v) */
        x /= 2; y /= 2;
    }
    return depth;
}
/* Scene::getDetphAtLevel, src/hpmvs/Scene.cpp:383-404 */
static float get_depth_at_level(const orc_depths *D, int img, int xx, int yy, int level) {
    const int x = sub_index(xx), y = sub_index(yy);
    const int k = img * ORC_MAX_LEVELS + level;
    if (x < 0 || x >= D->cols[k] || y < 0 || y >= D->rows[k]) return ORC_MAX_DEPTH;
    return D->d[k][y + (size_t)x * D->rows[k]];
}
/* Scene::depthTest(patch, ix, iy, depth, image, margin, viewBlock), src/hpmvs/Scene.cpp:552-580.
 * abs_int: the similarity test is written `abs(diff) < ...` on a float, unqualified, inside namespace mo3d
 * (Scene.cpp:571): with <cmath>'s overloads visible that is fabsf; with only C's abs(int) in scope the
 * difference is truncated to an int first.  Both are "the reference"; the caller says which. */
static int depth_test_px(const orc_scene *s, const orc_depths *D, const orc_patch *p, int ix, int iy, float depth,
                         int image, float margin, int view_block, int abs_int) {
    if (depth < 0 || ix < 0 || ix >= s->imgs[image].w[0] || iy < 0 || iy >= s->imgs[image].h[0]) return 0;
    const float img_depth = get_full_depth(s, D, image, ix, iy);
    if (img_depth >= ORC_MAX_DEPTH) return view_block ? 0 : 1;
    float d[4], ray[4];
    sub4(p->center, s->cams[image].center, d);
    normalized4(d, ray);
    const float diff = img_depth - depth;
    const float rn = dot4(ray, p->normal);
    const float factor = 2.0f < 2.0f + rn ? 2.0f : 2.0f + rn;  /* std::min(2.0f, 2.0f + ray.dot(normal)) */
    const double thr = p->scale * margin * factor * 2.0;
    if (!view_block) {
        const double a = abs_int ? (double)abs((int)diff) : (double)fabsf(diff);
        return a < thr;
    }
    return (double)diff > thr;
}
/* Scene::depthTest(patch, image, margin, neighbours = true, viewBlock), src/hpmvs/Scene.cpp:531-550 */
static int depth_test(const orc_scene *s, const orc_depths *D, const orc_patch *p, int image, float margin,
                      int view_block, int abs_int) {
    float c[3];
    cam_mult(&s->cams[image], p->center, 0, c);
    int ix = (int)(c[0] / c[2] + 0.5), iy = (int)(c[1] / c[2] + 0.5);
    ix--; iy--;
    for (int yy = 0; yy < 3; yy++)
        for (int xx = 0; xx < 3; xx++)
            if (!depth_test_px(s, D, p, ix + xx, iy + yy, c[2], image, margin, view_block, abs_int)) return 0;
    return 1;
}
/* Scene::pixelFreeTest, src/hpmvs/Scene.cpp:590-605 */
static int pixel_free_test(const orc_scene *s, const orc_depths *D, const orc_patch *p, int image) {
    const orc_camera *cam = &s->cams[image];
    const int level = (int)roundf(cam_get_level(cam, p->center, p->scale));
    if (level < 0 || level >= cam->n_levels) return 0;
    float c[3];
    cam_project(cam, p->center, level, c);
    const int ix = (int)(c[0] / c[2] + 0.5), iy = (int)(c[1] / c[2] + 0.5);
    if (ix < 0 || ix >= s->imgs[image].w[level] || iy < 0 || iy >= s->imgs[image].h[level]) return 0;
    return get_depth_at_level(D, image, ix, iy, level) == ORC_MAX_DEPTH;
}
/* Scene::depthTests / viewBlockTest / pixelFreeTests (src/hpmvs/Scene.cpp:518-524, 607-642, 582-588): the three
 * counts CellProcessor::extend / ::branch / ::refine gate a refined candidate on (CellProcessor.cpp:134-142). */
void orc_depth_gates(const orc_scene *s, const orc_depths *D, const orc_patch *p, float margin, int abs_int,
                     int *n_visible, int *n_blocking, int *n_free) {
    int vis = 0, blk = 0, fre = 0;
    for (int i = 0; i < p->n_images; i++) {
        if (depth_test(s, D, p, p->images[i], margin, 0, abs_int)) vis++;
        if (pixel_free_test(s, D, p, p->images[i])) fre++;
    }
    for (int img = 0; img < s->n_views; img++) {
        const orc_camera *cam = &s->cams[img];
        const int level = (int)roundf(cam_get_level(cam, p->center, p->scale));
        if (level < 0 || level > cam->n_levels - 1) continue;
        float c[3];
        cam_project(cam, p->center, level, c);
        if (c[0] < 0 || c[0] > s->imgs[img].w[level] || c[1] < 0 || c[1] > s->imgs[img].h[level]) continue;
        if (depth_test(s, D, p, img, margin, 1, abs_int)) blk++;
    }
    *n_visible = vis; *n_blocking = blk; *n_free = fre;
}

/* ------------------------------------------------------------------------- */
/* One priority level of the expansion, sequentially, as the reference runs it.
 *
 * main.cpp:146-181 pops the leaves of one priority value and calls CellProcessor::extend on each
 * (src/hpmvs/CellProcessor.cpp:84-178): six hexagon candidates at the leaf's width; a candidate whose octree leaf is
 * already occupied is skipped BEFORE optimize (:118-122); the refined candidate has to pass the scale and drift gates
 * (:130-133), the three depth-map counts (:134-142: depthTests >= MIN_IMAGES_PER_PATCH, viewBlockTest <
 * MIN_IMAGES_PER_PATCH, pixelFreeTests >= MIN_IMAGES_PER_PATCH - 1 and > 75 % of its images), and
 * DynOctTree::addConditional (doctree.h:397-419: the leaf at the REFINED centre must be empty); then its depths are
 * entered (:165, Scene::setDepths) and it joins the next priority level (:171-175).  Everything a later candidate reads
 * -- occupancy and depth maps -- already contains what the earlier ones of the same level wrote.
 *
 * The octree is replaced by a caller-owned occupancy set over the uniform grid of the level's leaf width (the leaves of
 * one priority level have one width; the root contains everything): the cell of a point is floor(p / width) per axis.
 *
 * frozen_gates != 0: the three counts read a copy of the depth maps taken when the round starts -- what a
 * frontier-batched round sees (every candidate of the level refined and gated in one launch) -- while
 * insertion, occupancy and setDepths stay ordered.  The difference between the two modes IS the interaction between the
 * candidates of one level.
 *
 * out[6 n]: every candidate as extend() left it (stage 0 accepted; 20 skipped before optimize; 1-9 / 100 optimize();
 * 21 scale gate; 22 drift gate; 23 depthTests; 24 viewBlockTest; 25 pixelFreeTests; 26 addConditional).
 * counts[3 * 6 n]: the three counts of the candidates that reached them (-1 otherwise).  Returns the number accepted. */
typedef struct { long long *keys; int n, cap; } orc_occ;
static long long occ_key(const float *p, float width) {
    const long long ix = (long long)floorf(p[0] / width), iy = (long long)floorf(p[1] / width), iz = (long long)floorf(p[2] / width);
    return ((ix + (1 << 20)) << 42) | ((iy + (1 << 20)) << 21) | (iz + (1 << 20));
}
static int occ_has(const orc_occ *o, long long k) {
    for (int i = 0; i < o->n; i++) if (o->keys[i] == k) return 1;
    return 0;
}
long long orc_cell_key(const float *p, float width) { return occ_key(p, width); }

int orc_extend_round(const orc_scene *s, const orc_options *o, orc_depths *D, int n_leaves, const orc_patch *parents,
                     float width, float margin, int abs_int, long long *occ_keys, int *n_occ, int occ_cap,
                     int frozen_gates, orc_patch *out, int *counts, int which) {
    orc_occ occ = {occ_keys, *n_occ, occ_cap};
    orc_depths *G = D;  /* the maps the gates read */
    if (frozen_gates) {
        G = orc_depths_create(s);
        for (int k = 0; k < D->n_views * ORC_MAX_LEVELS; k++)
            if (D->d[k]) memcpy(G->d[k], D->d[k], sizeof(float) * (size_t)D->rows[k] * D->cols[k]);
    }
    int accepted = 0;
    const float cc[3] = {0, 0, 0};
    for (int i = 0; i < n_leaves; i++) {
        for (int ii = 0; ii < 6; ii++) {
            unsigned char skip[6] = {1, 1, 1, 1, 1, 1};
            orc_patch six[6];
            int *cnt = counts + 3 * (6 * i + ii);
            cnt[0] = cnt[1] = cnt[2] = -1;
            /* the candidate before optimize: its leaf must be empty (CellProcessor.cpp:118-122) */
            orc_expand(s, o, 0, &parents[i], cc, width, skip, six, which);  /* (constructs; optimizes nothing) */
            orc_patch *q = &out[6 * i + ii];
            if (occ_has(&occ, occ_key(six[ii].center, width))) { *q = six[ii]; q->stage = 20; continue; }
            skip[ii] = 0;
            orc_expand(s, o, 0, &parents[i], cc, width, skip, six, which);  /* optimize() + the scale and drift gates */
            *q = six[ii];
            if (q->stage != 0) continue;
            int v, b, f;
            orc_depth_gates(s, G, q, margin, abs_int, &v, &b, &f);
            cnt[0] = v; cnt[1] = b; cnt[2] = f;
            if (!(v >= o->MIN_IMAGES_PER_PATCH)) { q->stage = 23; continue; }
            if (!(b < o->MIN_IMAGES_PER_PATCH)) { q->stage = 24; continue; }
            if (!(f >= o->MIN_IMAGES_PER_PATCH - 1 && f * 1.0 / q->n_images > 0.75)) { q->stage = 25; continue; }
            const long long key = occ_key(q->center, width);
            if (occ_has(&occ, key) || occ.n >= occ.cap) { q->stage = 26; continue; }  /* addConditional */
            occ.keys[occ.n++] = key;
            orc_set_depths(s, D, q);
            accepted++;
        }
    }
    if (frozen_gates) orc_depths_destroy(G);
    *n_occ = occ.n;
    return accepted;
}

/* One priority level of CellProcessor::branch (src/hpmvs/CellProcessor.cpp:210-307), leaf by leaf in order.  Per leaf: its
 * patch must have level support (:221-224), the four diagonal children are built, gated by Cell::contains before and after
 * optimize (:247-258: orc_expand, mode 1); a leaf on the final tree level that got no child keeps its patch (:265-266,
 * `final_level[i]` = nodeLevel(cell) >= PATCH_FINAL_MINLEVEL, the scheduler's knowledge); otherwise the leaf is split: its
 * patch's depths are taken back (:276-279) and the children's are entered (:296).  The octree itself (split, the new leaves,
 * the queue) stays with the caller.  out: 4 candidates per leaf; split[i]: 1 if leaf i was split.  Returns the split leaves. */
int orc_branch_round(const orc_scene *s, const orc_options *o, orc_depths *D, int n_leaves, const orc_patch *parents,
                     const float *cell_center, const float *cell_width, const unsigned char *final_level, orc_patch *out,
                     unsigned char *split, int which) {
    int n_split = 0;
    for (int i = 0; i < n_leaves; i++) {
        orc_patch *four = out + (size_t)4 * i;
        split[i] = 0;
        if (orc_level_support(s, &parents[i], o->MINLEVEL) < 1) {   /* exhausted: nothing is built */
            const unsigned char all[4] = {1, 1, 1, 1};
            orc_expand(s, o, 1, &parents[i], cell_center + 3 * i, cell_width[i], all, four, which);
            for (int k = 0; k < 4; k++) four[k].stage = 27;
            continue;
        }
        orc_expand(s, o, 1, &parents[i], cell_center + 3 * i, cell_width[i], NULL, four, which);
        int children = 0;
        for (int k = 0; k < 4; k++) children += four[k].stage == 0 ? 1 : 0;
        if (final_level[i] && children == 0) continue;
        split[i] = 1; n_split++;
        orc_set_depths_ex(s, D, &parents[i], 1);
        for (int k = 0; k < 4; k++) if (four[k].stage == 0) orc_set_depths_ex(s, D, &four[k], 0);
    }
    return n_split;
}
