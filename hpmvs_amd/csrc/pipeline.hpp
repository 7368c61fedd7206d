// pipeline.hpp -- the image-list passes of PatchOptimizer::runOptimization (reference
// src/hpmvs/PatchOptimizer.cpp:105-284) and Scene::getColor(patch) (src/hpmvs/Scene.cpp:300-327) as
// WAVE-COOPERATIVE device code: one request (one patch's list) at a time, lane k standing for list entry k
// (or for covisible view k of a 64-wide chunk).
//
// Rounds 1-2 ran these passes as per-lane scalar code inside phase A of the refinement kernel: addImages walks up
// to 49 covisible views serially (float64 log2, three divisions and ~30 scattered loads per view), sortImages is
// O(m^2) over scratch-memory arrays, assureImageAngles evaluates up to m^2/2 acosf -- 87 k wave-cycles per call
// while the other lanes of the wavefront waited: 14 % of the kernel at BASELINE configs[2]
// (profiles/r03/phase_c2_rows.txt).  Here the per-entry work of a pass runs on 64 lanes at once and the
// order-dependent parts (append order, greedy selection, stable compaction) are ballots, ranks and uniform
// loops.  Every float operation per entry is the reference's, in the reference's order, so the resulting
// lists are identical.
//
// All functions must be called by all 64 lanes with wave-uniform arguments (the caller reads the requesting
// lane's values with v_readlane); `id` is the register copy of the list: lane k < nimg holds image id k.
#pragma once
#include <hip/hip_runtime.h>

#include "dev_types.h"
#include "photometric.hpp"

namespace hpmvs {

namespace {

enum Op { OP_NONE = 0, OP_OBJ = 1, OP_INCC = 2, OP_REFSUM = 3,
          OPL_ADD = 4, OPL_F1 = 5, OPL_F2 = 6, OPL_F3 = 7,  // list services (this file)
          OPQ_START = 8, OPQ_STEP = 9 };                    // optimiser service (kernel_optimize.hip)
enum Stage {
    S_FETCH = 0, S_WAIT, S_ADD1, S_F1, S_F1B, S_BQ, S_BQFAIL, S_ADD2, S_F2, S_F2B, S_RS, S_F3, S_F3B, S_IDLE
};

struct U3 { float x, y, z; };  // a wave-uniform 3-vector

// What the list services need of DevScene / DevOptions, and nothing else: their arguments travel in vector registers (a real
// call), and the two structures by value were 38 of them at every call site of a kernel that has none to spare.
struct LevelThr { double t[kMaxLevels]; };   // DevScene::lvl_thr by value
struct ListEnv {
    const DevView* views; const int32_t* covis_off; const int32_t* covis;
    float cos_max_f, sort_thr, MAX_ANGLE, MIN_ANGLE;
    int32_t MAXLEVEL, MINLEVEL, MIN_IMAGES_PER_PATCH;
};
__device__ __forceinline__ ListEnv list_env(const DevScene& sc, const DevOptions& o) {
    return ListEnv{sc.views, sc.covis_off, sc.covis, o.cos_max_f, o.sort_thr, o.MAX_ANGLE, o.MIN_ANGLE, o.MAXLEVEL, o.MINLEVEL, o.MIN_IMAGES_PER_PATCH};
}

__device__ __forceinline__ void ray_to_cam(const DevView& V, const float* c, float* ray) {
    // (camera.center_ - pCenter).normalized() on 4-vectors whose w components cancel
    const float d[3] = {V.center[0] - c[0], V.center[1] - c[1], V.center[2] - c[2]};
    normalized3f(d, ray);
}

__device__ __forceinline__ unsigned long long lanes_below(int lane) { return (1ull << lane) - 1ull; }

// stable compaction of the register list: entry k survives iff bit k of `keep`; returns the new count
__device__ __forceinline__ int compact_list(int& id, unsigned long long keep) {
    const int lane = lane_id();
    int nid = -1, m = 0;
    for (unsigned long long t = keep; t; t &= t - 1ull, m++) {
        const int v = bcast_i(id, __ffsll((long long)t) - 1);
        if (lane == m) nid = v;
    }
    id = nid;
    return m;
}

// minimum over the lanes of `v` (lanes that do not take part pass +inf).  DPP: four shifts inside the rows of 16 lanes, the
// rows' results handed on with row_bcast, the total read from lane 63 -- seven dependent VALU operations where six
// __shfl_xor steps are six ds_bpermute round trips (sortImages' greedy selection calls this once per pick).  min is
// associative and commutative and v_min_f32 treats NaN the same way in either order, so the value is the same.
__device__ __forceinline__ float wave_min_f32(float v) {
#define HPMVS_MIN_DPP(ctrl, rows) v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, rows, 0xf, false)))
    HPMVS_MIN_DPP(0x111, 0xf);  // row_shr:1 (a lane without a left neighbour in its row reads itself)
    HPMVS_MIN_DPP(0x112, 0xf);
    HPMVS_MIN_DPP(0x114, 0xf);
    HPMVS_MIN_DPP(0x118, 0xf);
    HPMVS_MIN_DPP(0x142, 0xa);  // row_bcast:15 -> rows 1 and 3
    HPMVS_MIN_DPP(0x143, 0xc);  // row_bcast:31 -> rows 2 and 3
#undef HPMVS_MIN_DPP
    return bcast_f(v, 63);
}

// PatchOptimizer::addImages (PatchOptimizer.cpp:225-258): the test one covisible view has to pass
template <class OP>
__device__ __forceinline__ bool add_images_test(const DevView& V, const OP& o, const float* c, const float* n, float scale) {
    float ray[3];
    ray_to_cam(V, c, ray);
    if (dot3f(ray, n) < o.cos_max_f) return false;
    // Camera::getLevel (Camera.cpp:92-95) then std::round
    const float dd[3] = {c[0] - V.center[0], c[1] - V.center[1], c[2] - V.center[2]};
    const float fz = sqrtf(dot3f(dd, dd));
    const float lv = (float)log2((double)(scale * V.fsum) / (2.0 * (double)fz));
    const int il = cvt_i32_x86(roundf(lv));
    if (il < o.MINLEVEL || il >= o.MAXLEVEL - 2) return false;
    float P[12];
    for (int k = 0; k < 12; k++) P[k] = V.P0[k];
    float u, v;
    project0(P, c[0], c[1], c[2], u, v);
    const float s = __int_as_float((127 - il) << 23);
    u *= s; v *= s;
    if (u < 0.0f || (float)(V.w[il] - 1) <= u || v < 0.0f || (float)(V.h[il] - 1) <= v) return false;
    return true;
}

// addImages on the list at `row` (global workspace, nimg entries): the covisible views of image 0 that are not
// attached yet and pass the test are appended in covisibility order.  Returns the new count, -1 on overflow.
__device__ __noinline__ int coop_add_images(ListEnv sc, uint16_t* row, int nimg, U3 c_, U3 n_, float scale) {
    if (nimg <= 0) return 0;
    const int lane = lane_id();
    const float c[3] = {c_.x, c_.y, c_.z}, n[3] = {n_.x, n_.y, n_.z};
    const int id = (lane < nimg) ? (int)row[lane] : -1;
    const int ref = bcast_i(id, 0);
    const int existing = nimg;
    const int beg = sc.covis_off[ref], end = sc.covis_off[ref + 1];
    for (int base = beg; base < end; base += 64) {
        const int ci = base + lane;
        const bool valid = ci < end;
        const int cv = valid ? sc.covis[ci] : -2;
        bool found = false;
        for (int k = 0; k < existing; k++) found |= (bcast_i(id, k) == cv);
        bool pass = false;
        if (valid && !found) pass = add_images_test(sc.views[cv], sc, c, n, scale);
        const unsigned long long m = __ballot(pass);
        const int cnt = __popcll(m);
        if (nimg + cnt > kMaxImages) return -1;
        if (pass) row[nimg + __popcll(m & lanes_below(lane))] = (uint16_t)cv;
        nimg += cnt;
    }
    return nimg;
}

// PatchOptimizer::sortImages + getAngleWeightedScales (PatchOptimizer.cpp:183-223, 260-284) on the register
// list; returns the new count (the reference ignores the boolean result).
template <class SC, class OP>
__device__ __forceinline__ int coop_sort_images(const SC& sc, const OP& o, int& id, int nimg, const float* c,
                                                const float* n, float scale) {
    const int lane = lane_id();
    bool cand = false;
    float w = 0.0f, ray[3] = {0.0f, 0.0f, 0.0f};
    if (nimg > 0) {
        const DevView& V0 = sc.views[bcast_i(id, 0)];
        const float dd[3] = {c[0] - V0.center[0], c[1] - V0.center[1], c[2] - V0.center[2]};
        const float fz0 = sqrtf(dot3f(dd, dd));
        int refLevel = cvt_i32_x86(roundf((float)log2((double)(scale * V0.fsum) / (2.0 * (double)fz0))));
        refLevel = refLevel < o.MAXLEVEL - 1 ? refLevel : o.MAXLEVEL - 1;
        refLevel = refLevel > 0 ? refLevel : 0;
        float nn[3];
        normalized3f(n, nn);
        if (lane < nimg) {
            const DevView& V = sc.views[id];
            ray_to_cam(V, c, ray);
            const float cosa = dot3f(ray, nn);
            if (cosa > 0.0f) {
                cand = true;
                // Camera::getScale (Camera.cpp:83-90)
                const float e[3] = {c[0] - V.center[0], c[1] - V.center[1], c[2] - V.center[2]};
                const float fz = sqrtf(dot3f(e, e));
                const float sc_ = (V.fsum == 0.0f) ? 1.0f : (float)(2.0 * (double)fz * (double)(1 << refLevel) / (double)V.fsum);
                w = sc_ / cosa;
            }
        }
    }
    unsigned long long alive = __ballot(cand);
    if (__popcll(alive) < 2) { id = -1; return 0; }  // pImages_ was cleared
    if (lane == __ffsll((long long)alive) - 1) w = 0.0f;  // the first candidate keeps its place
    const float thr = o.sort_thr;
    int out = 0, nid = -1;
    while (alive) {
        // greedy pick: the smallest weight, the earliest entry among equals
        const bool mine = (alive >> lane) & 1ull;
        const float mn = wave_min_f32(mine ? w : __int_as_float(0x7f800000));
        // (weights that are all NaN -- a camera with a non-finite focal sum -- compare equal to nothing: the earliest
        //  remaining entry then, so that every pass removes one entry and the loop ends)
        const unsigned long long eq = __ballot(mine && w == mn);
        const int best = __ffsll((long long)(eq ? eq : alive)) - 1;
        const int bid = bcast_i(id, best);
        if (lane == out) nid = bid;
        out++;
        const float rb[3] = {bcast_f(ray[0], best), bcast_f(ray[1], best), bcast_f(ray[2], best)};
        alive &= ~(1ull << best);
        if ((alive >> lane) & 1ull) {
            const float ftmp = fminf(thr, fmaxf(thr / 2.0f, 1.0f - dot3f(rb, ray)));
            w = w * (thr / ftmp);
        }
    }
    id = nid;
    return out;
}

// PatchOptimizer::assureImageAngles (PatchOptimizer.cpp:105-123)
template <class SC, class OP>
__device__ __forceinline__ bool coop_assure_image_angles(const SC& sc, const OP& o, int id, int nimg, const float* c) {
    const int lane = lane_id();
    float rj[3] = {0.0f, 0.0f, 0.0f};
    if (lane < nimg) ray_to_cam(sc.views[id], c, rj);
    for (int ii = 0; ii < nimg - 1; ii++) {
        const float ri[3] = {bcast_f(rj[0], ii), bcast_f(rj[1], ii), bcast_f(rj[2], ii)};
        bool hit = false;
        if (lane > ii && lane < nimg) {
            const float a = acosf(dot3f(ri, rj));
            hit = a < o.MAX_ANGLE && a > o.MIN_ANGLE;
        }
        if (__any(hit)) return true;
    }
    return false;
}

// PatchOptimizer::filterImagesByAngle (PatchOptimizer.cpp:125-136)
template <class SC, class OP>
__device__ __forceinline__ int coop_filter_images_by_angle(const SC& sc, const OP& o, int& id, int nimg,
                                                           const float* c, const float* n) {
    bool keep = false;
    if (lane_id() < nimg) {
        float ray[3];
        ray_to_cam(sc.views[id], c, ray);
        keep = dot3f(ray, n) > o.cos_max_f;
    }
    return compact_list(id, __ballot(keep));
}

// Scene::getColor(const Patch3d&) (Scene.cpp:300-327): bilinear colour at the projected centre in
// every attached image, median by norm (darkest if the median is brighter than 250).
__device__ __forceinline__ void colour_at(const DevScene& sc, const DevView& V, const float* c, float scale, float* rgb) {
    const float dd[3] = {c[0] - V.center[0], c[1] - V.center[1], c[2] - V.center[2]};
    const float fz = sqrtf(dot3f(dd, dd));
    const int lvl = level_clamped(sc, scale, V.fsum, fz, V.n_levels - 1);
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

__device__ __forceinline__ U3 coop_patch_colour(const DevScene& sc, int id, int nimg, const float* c, float scale) {
    const int lane = lane_id();
    float rgb[3] = {0.0f, 0.0f, 0.0f}, norm = 0.0f;
    if (lane < nimg) {
        colour_at(sc, sc.views[id], c, scale, rgb);
        norm = sqrtf(dot3f(rgb, rgb));
    }
    // rank selection instead of std::sort: element of rank nimg/2 and the minimum
    int rank = 0;
    for (int j = 0; j < nimg; j++) {
        const float nj = bcast_f(norm, j);
        rank += (nj < norm) || (nj == norm && j < lane);
    }
    const int want = nimg / 2;
    const unsigned long long mm = __ballot(lane < nimg && rank == want), m0 = __ballot(lane < nimg && rank == 0);
    const int med = mm ? __ffsll((long long)mm) - 1 : 0, mn = m0 ? __ffsll((long long)m0) - 1 : 0;
    const int pick = (bcast_f(norm, med) > 250.0f) ? mn : med;
    return U3{bcast_f(rgb[0], pick), bcast_f(rgb[1], pick), bcast_f(rgb[2], pick)};
}

// ---- the three list services between the photometric requests of runOptimization (PatchOptimizer.cpp:48-76)
struct ListResult { int nimg; int fail; U3 rgb; };

__device__ __forceinline__ void store_list(uint16_t* row, int id, int nimg) {
    if (lane_id() < nimg) row[lane_id()] = (uint16_t)id;
}

// filterImagesNCC(alpha1)'s removal -> sortImages -> assureImageAngles (PatchOptimizer.cpp:52-56, 144-151)
__device__ __noinline__ ListResult coop_list_f1(ListEnv sc, uint16_t* row, int nimg, unsigned long long keep,
                                               U3 c_, U3 n_, float scale) {
    const ListEnv& o = sc;
    const int lane = lane_id();
    const float c[3] = {c_.x, c_.y, c_.z}, n[3] = {n_.x, n_.y, n_.z};
    ListResult r;
    r.fail = 0; r.rgb = U3{0.0f, 0.0f, 0.0f};
    int id = (lane < nimg) ? (int)row[lane] : -1;
    const unsigned long long all = nimg >= 64 ? ~0ull : ((1ull << nimg) - 1ull);
    nimg = compact_list(id, (keep | 1ull) & all);  // image 0 always stays
    r.nimg = nimg;
    if (nimg < o.MIN_IMAGES_PER_PATCH) { r.fail = 2; return r; }
    nimg = coop_sort_images(sc, o, id, nimg, c, n, scale);
    r.nimg = nimg;
    if (!coop_assure_image_angles(sc, o, id, nimg, c)) { r.fail = 3; return r; }
    if (nimg < o.MIN_IMAGES_PER_PATCH) { r.fail = 4; return r; }
    store_list(row, id, nimg);
    return r;
}

// filterImagesNCC(alpha2)'s removal -> filterImagesByAngle -> assureImageAngles (PatchOptimizer.cpp:62-67)
__device__ __noinline__ ListResult coop_list_f2(ListEnv sc, uint16_t* row, int nimg, unsigned long long keep,
                                               U3 c_, U3 n_) {
    const ListEnv& o = sc;
    const int lane = lane_id();
    const float c[3] = {c_.x, c_.y, c_.z}, n[3] = {n_.x, n_.y, n_.z};
    ListResult r;
    r.fail = 0; r.rgb = U3{0.0f, 0.0f, 0.0f};
    int id = (lane < nimg) ? (int)row[lane] : -1;
    const unsigned long long all = nimg >= 64 ? ~0ull : ((1ull << nimg) - 1ull);
    nimg = compact_list(id, (keep | 1ull) & all);
    r.nimg = nimg;
    if (nimg < o.MIN_IMAGES_PER_PATCH) { r.fail = 6; return r; }
    nimg = coop_filter_images_by_angle(sc, o, id, nimg, c, n);
    r.nimg = nimg;
    if (nimg < o.MIN_IMAGES_PER_PATCH) { r.fail = 7; return r; }
    if (!coop_assure_image_angles(sc, o, id, nimg, c)) { r.fail = 8; return r; }
    store_list(row, id, nimg);
    return r;
}

// the last filterImagesNCC(alpha2)'s removal (PatchOptimizer.cpp:73) and Scene::getColor(patch) (PatchOptimizer.cpp:100)
__device__ __noinline__ ListResult coop_list_f3(const DevView* views, LevelThr lt, int min_images, uint16_t* row, int nimg, unsigned long long keep,
                                               U3 c_, float scale, int max_images, int want_colour) {
    DevScene sc;  // (what colour_at / level_clamped read of it)
    sc.views = views; sc.covis_off = nullptr; sc.covis = nullptr; sc.n_views = 0;
    for (int k = 0; k < kMaxLevels; k++) sc.lvl_thr[k] = lt.t[k];
    const int lane = lane_id();
    const float c[3] = {c_.x, c_.y, c_.z};
    ListResult r;
    r.fail = 0; r.rgb = U3{0.0f, 0.0f, 0.0f};
    int id = (lane < nimg) ? (int)row[lane] : -1;
    const unsigned long long all = nimg >= 64 ? ~0ull : ((1ull << nimg) - 1ull);
    nimg = compact_list(id, (keep | 1ull) & all);
    r.nimg = nimg;
    if (nimg < min_images) { r.fail = 9; return r; }
    if (nimg > max_images) { r.fail = 100; return r; }
    store_list(row, id, nimg);
    if (want_colour) r.rgb = coop_patch_colour(sc, id, nimg, c, scale);
    return r;
}

}  // namespace

}  // namespace hpmvs
