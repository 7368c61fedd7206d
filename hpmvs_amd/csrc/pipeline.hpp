// pipeline.hpp -- per-lane scalar device code of PatchOptimizer::runOptimization's image-list passes
// (reference src/hpmvs/PatchOptimizer.cpp:105-284) and Scene::getColor(patch) (src/hpmvs/Scene.cpp:300-327),
// shared by the two refinement kernels (kernel_optimize.hip: lane-bound patches, low latency;
// kernel_stream.hip: patches streaming through per-block queues, high throughput).
#pragma once
#include <hip/hip_runtime.h>

#include "dev_types.h"
#include "photometric.hpp"

namespace hpmvs {

namespace {

enum Op { OP_NONE = 0, OP_OBJ = 1, OP_INCC = 2, OP_REFSUM = 3 };
enum Stage {
    S_FETCH = 0, S_F1, S_BQ, S_F2, S_RS, S_F3, S_IDLE
};

__device__ __forceinline__ void ray_to_cam(const DevView& V, const float* c, float* ray) {
    // (camera.center_ - pCenter).normalized() on 4-vectors whose w components cancel
    const float d[3] = {V.center[0] - c[0], V.center[1] - c[1], V.center[2] - c[2]};
    normalized3f(d, ray);
}

// PatchOptimizer::addImages (PatchOptimizer.cpp:225-258).  Returns the new count, or -1 on overflow.
__device__ __noinline__ int add_images(const DevScene& sc, const DevOptions& o, uint16_t* row, int nimg, const float* c,
                          const float* n, float scale) {
    if (nimg <= 0) return 0;
    const int ref = row[0];
    const int existing = nimg;
    const int beg = sc.covis_off[ref], end = sc.covis_off[ref + 1];
    for (int ci = beg; ci < end; ci++) {
        const int cv = sc.covis[ci];
        bool found = false;
        for (int k = 0; k < existing; k++) found |= (row[k] == cv);
        if (found) continue;
        const DevView& V = sc.views[cv];
        float ray[3];
        ray_to_cam(V, c, ray);
        if (dot3f(ray, n) < o.cos_max_f) continue;
        // Camera::getLevel (Camera.cpp:92-95) then std::round
        const float dd[3] = {c[0] - V.center[0], c[1] - V.center[1], c[2] - V.center[2]};
        const float fz = sqrtf(dot3f(dd, dd));
        const float lv = (float)log2((double)(scale * V.fsum) / (2.0 * (double)fz));
        const int il = (int)roundf(lv);
        if (il < o.MINLEVEL || il >= o.MAXLEVEL - 2) continue;
        float P[12];
        for (int k = 0; k < 12; k++) P[k] = V.P0[k];
        float u, v;
        project0(P, c[0], c[1], c[2], u, v);
        const float s = __int_as_float((127 - il) << 23);
        u *= s; v *= s;
        if (u < 0.0f || (float)(V.w[il] - 1) <= u || v < 0.0f || (float)(V.h[il] - 1) <= v) continue;
        if (nimg >= kMaxImages) return -1;
        row[nimg++] = (uint16_t)cv;
    }
    return nimg;
}

// PatchOptimizer::sortImages + getAngleWeightedScales (PatchOptimizer.cpp:183-223, 260-284).
// Rewrites the row; returns the new count (the reference ignores the boolean result).
__device__ __noinline__ int sort_images(const DevScene& sc, const DevOptions& o, uint16_t* row, int nimg, const float* c,
                           const float* n, float scale) {
    uint16_t idx[kMaxImages];
    float ws[kMaxImages];
    float rays[kMaxImages][3];
    int m = 0;
    if (nimg > 0) {
        const DevView& V0 = sc.views[row[0]];
        const float dd[3] = {c[0] - V0.center[0], c[1] - V0.center[1], c[2] - V0.center[2]};
        const float fz0 = sqrtf(dot3f(dd, dd));
        int refLevel = (int)roundf((float)log2((double)(scale * V0.fsum) / (2.0 * (double)fz0)));
        refLevel = refLevel < o.MAXLEVEL - 1 ? refLevel : o.MAXLEVEL - 1;
        refLevel = refLevel > 0 ? refLevel : 0;
        float nn[3];
        normalized3f(n, nn);
        for (int k = 0; k < nimg; k++) {
            const DevView& V = sc.views[row[k]];
            float ray[3];
            ray_to_cam(V, c, ray);
            const float cosa = dot3f(ray, nn);
            if (cosa > 0.0f) {
                idx[m] = row[k];
                rays[m][0] = ray[0]; rays[m][1] = ray[1]; rays[m][2] = ray[2];
                // Camera::getScale (Camera.cpp:83-90)
                const float e[3] = {c[0] - V.center[0], c[1] - V.center[1], c[2] - V.center[2]};
                const float fz = sqrtf(dot3f(e, e));
                const float sc_ = (V.fsum == 0.0f) ? 1.0f : (float)(2.0 * (double)fz * (double)(1 << refLevel) / (double)V.fsum);
                ws[m] = sc_ / cosa;
                m++;
            }
        }
    }
    if (m < 2) return 0;  // pImages_ was cleared
    ws[0] = 0.0f;
    const float thr = o.sort_thr;
    int out = 0;
    while (m > 0) {
        int best = 0;
        for (int j = 1; j < m; j++)
            if (ws[j] < ws[best]) best = j;
        row[out++] = idx[best];
        const float rb[3] = {rays[best][0], rays[best][1], rays[best][2]};
        int q = 0;
        for (int j = 0; j < m; j++) {
            if (j == best) continue;
            const float ftmp = fminf(thr, fmaxf(thr / 2.0f, 1.0f - dot3f(rb, rays[j])));
            const float w = ws[j] * (thr / ftmp);
            idx[q] = idx[j];
            rays[q][0] = rays[j][0]; rays[q][1] = rays[j][1]; rays[q][2] = rays[j][2];
            ws[q] = w;
            q++;
        }
        m = q;
    }
    return out;
}

// PatchOptimizer::assureImageAngles (PatchOptimizer.cpp:105-123)
__device__ __noinline__ bool assure_image_angles(const DevScene& sc, const DevOptions& o, const uint16_t* row, int nimg,
                                    const float* c) {
    for (int ii = 0; ii < nimg - 1; ii++) {
        float ri[3];
        ray_to_cam(sc.views[row[ii]], c, ri);
        for (int jj = ii + 1; jj < nimg; jj++) {
            float rj[3];
            ray_to_cam(sc.views[row[jj]], c, rj);
            const float a = acosf(dot3f(ri, rj));
            if (a < o.MAX_ANGLE && a > o.MIN_ANGLE) return true;
        }
    }
    return false;
}

// PatchOptimizer::filterImagesByAngle (PatchOptimizer.cpp:125-136)
__device__ __noinline__ int filter_images_by_angle(const DevScene& sc, const DevOptions& o, uint16_t* row, int nimg,
                                      const float* c, const float* n) {
    int m = 0;
    for (int k = 0; k < nimg; k++) {
        float ray[3];
        ray_to_cam(sc.views[row[k]], c, ray);
        if (dot3f(ray, n) > o.cos_max_f) row[m++] = row[k];
    }
    return m;
}

// keep image 0 and the images whose bit is set (PatchOptimizer.cpp:144-151)
__device__ int apply_keep_mask(uint16_t* row, int nimg, unsigned long long keep) {
    int m = 1;
    for (int i = 1; i < nimg; i++)
        if ((keep >> i) & 1ull) row[m++] = row[i];
    return m;
}

// Scene::getColor(const Patch3d&) (Scene.cpp:300-327): bilinear colour at the projected centre in
// every attached image, median by norm (darkest if the median is brighter than 250).
__device__ void colour_at(const DevView& V, const float* c, float scale, float* rgb) {
    const float dd[3] = {c[0] - V.center[0], c[1] - V.center[1], c[2] - V.center[2]};
    const float fz = sqrtf(dot3f(dd, dd));
    const int lvl = level_clamped(scale, V.fsum, fz, V.n_levels - 1);
    float P[12];
    for (int k = 0; k < 12; k++) P[k] = V.P0[k];
    float x, y;
    project0(P, c[0], c[1], c[2], x, y);
    const float s = __int_as_float((127 - lvl) << 23);
    x *= s; y *= s;
    const int W = V.w[lvl];
    const int lx = (int)x, ly = (int)y;
    const uint8_t* p0 = V.pix[lvl] + 3 * (ly * W + lx);
    const uint8_t* p1 = p0 + 3 * W;
    const float dx1 = x - (float)lx, dx0 = 1.0f - dx1;
    const float dy1 = y - (float)ly, dy0 = 1.0f - dy1;
    const float f00 = dx0 * dy0, f01 = dx0 * dy1, f10 = dx1 * dy0, f11 = dx1 * dy1;
    for (int ch = 0; ch < 3; ch++) {
        float v = (float)p0[ch] * f00 + (float)p1[ch] * f01;
        v += (float)p0[3 + ch] * f10 + (float)p1[3 + ch] * f11;
        rgb[ch] = v;
    }
}

__device__ __noinline__ void patch_colour(const DevScene& sc, const uint16_t* row, int nimg, const float* c, float scale,
                             float* out) {
    float norms[kMaxImages];
    for (int k = 0; k < nimg; k++) {
        float rgb[3];
        colour_at(sc.views[row[k]], c, scale, rgb);
        norms[k] = sqrtf(dot3f(rgb, rgb));
    }
    // rank selection instead of std::sort: element of rank nimg/2 and the minimum
    const int want = nimg / 2;
    int med = 0, mn = 0;
    for (int k = 0; k < nimg; k++) {
        int rank = 0;
        for (int j = 0; j < nimg; j++) rank += (norms[j] < norms[k]) || (norms[j] == norms[k] && j < k);
        if (rank == want) med = k;
        if (rank == 0) mn = k;
    }
    const int pick = (norms[med] > 250.0f) ? mn : med;
    colour_at(sc.views[row[pick]], c, scale, out);
}

}  // namespace

// The out-of-line helpers take pointers; handing them the kernel's long-lived per-lane arrays would pin those
// arrays in scratch memory for the whole kernel (an escaping address defeats register promotion, and every
// later c[k] / n[k] becomes a scratch access of several hundred cycles).  Short-lived copies escape instead;
// the same holds for the kernel arguments sc / o, which are copied per call (by-value parameters) so that the
// kernel's own copies stay in scalar registers.
struct F3 { float x, y, z; };
#define HPMVS_F3(a) F3{(a)[0], (a)[1], (a)[2]}
__device__ __forceinline__ int add_images_v(DevScene sc, DevOptions o, uint16_t* row, int nimg, F3 c, F3 n, float scale) {
    const float cc[3] = {c.x, c.y, c.z}, nn[3] = {n.x, n.y, n.z};
    return add_images(sc, o, row, nimg, cc, nn, scale);
}
__device__ __forceinline__ int sort_images_v(DevScene sc, DevOptions o, uint16_t* row, int nimg, F3 c, F3 n, float scale) {
    const float cc[3] = {c.x, c.y, c.z}, nn[3] = {n.x, n.y, n.z};
    return sort_images(sc, o, row, nimg, cc, nn, scale);
}
__device__ __forceinline__ bool assure_image_angles_v(DevScene sc, DevOptions o, const uint16_t* row, int nimg, F3 c) {
    const float cc[3] = {c.x, c.y, c.z};
    return assure_image_angles(sc, o, row, nimg, cc);
}
__device__ __forceinline__ int filter_images_by_angle_v(DevScene sc, DevOptions o, uint16_t* row, int nimg, F3 c, F3 n) {
    const float cc[3] = {c.x, c.y, c.z}, nn[3] = {n.x, n.y, n.z};
    return filter_images_by_angle(sc, o, row, nimg, cc, nn);
}
__device__ __forceinline__ F3 patch_colour_v(DevScene sc, const uint16_t* row, int nimg, F3 c, float scale) {
    const float cc[3] = {c.x, c.y, c.z};
    float rgb[3];
    patch_colour(sc, row, nimg, cc, scale, rgb);
    return F3{rgb[0], rgb[1], rgb[2]};
}

}  // namespace hpmvs
