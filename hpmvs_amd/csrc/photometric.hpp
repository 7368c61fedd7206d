// photometric.hpp -- gfx950 device code for the photometric primitives of the HPMVS hot path.
//
// Two kinds of code live here:
//   * per-lane scalar helpers (patch axes, optimiser-variable <-> centre/normal maps, the
//     per-image gates/level/projection "prep"): plain SIMT code, one patch or one image per lane;
//   * the wave-cooperative evaluator `wave_eval_exact`: ONE wavefront evaluates ONE patch against
//     its whole image list.  Lane v < nimg prepares image v (gates, pyramid level, projected window
//     geometry); per chunk of <= 8 images lanes 0..48 fetch the 7x7 window texels (lane = yy*7+xx,
//     two 8-byte loads per lane covering the 2x2x3 bilinear taps, all loads of the chunk in flight
//     together); means, variance and the NCC dot product are formed in the REFERENCE'S SEQUENTIAL
//     float32 summation order by single lanes (one per image / channel) reading LDS-staged values,
//     all images of the chunk in parallel.  Every float operation is the one the CPU reference
//     executes (the TU is built with -ffp-contract=off), so objective values are bit-identical.
//
// Reference functions restated here (file:line relative to the reference root):
//   calc_patch_axes        <- PatchOptimizer::calculatePatchAxis   src/hpmvs/PatchOptimizer.cpp:532-548
//   load_view, prep_from_view,
//   texel                  <- PatchOptimizer::sampleTexture        src/hpmvs/PatchOptimizer.cpp:476-529
//                             Camera::project / getLeveli          include/hpmvs/Camera.h:45-62, src/hpmvs/Camera.cpp:92-99
//                             Image::getColor                      include/hpmvs/Image.h:89-115
//   seq_sum + the staging  <- Patch2d::normalize / dot             include/hpmvs/Patch2d.hpp:37-84
//   wave_eval_exact(obj)   <- PatchOptimizer::objective_fn         src/hpmvs/PatchOptimizer.cpp:286-311
//   wave_eval_exact(!obj)  <- PatchOptimizer::setINCCs             src/hpmvs/PatchOptimizer.cpp:448-474
//   set_optimization_fields, set_center_norm, parameters_from_center_norm
//                          <- src/hpmvs/PatchOptimizer.cpp:384-446
//   asinf_glibc            <- glibc 2.35 __ieee754_asinf (what std::asin(float) is on the host)
#pragma once
#include <hip/hip_runtime.h>

#include "cr_acos.hpp"
#include "dev_types.h"

#include "prof.h"
#if defined(HPMVS_PROFILE_PHASES) && defined(__HIP_DEVICE_COMPILE__)
#define PH_T(v) const long long v = clock64()
#define PH_ADD(v0, v1, i) do { if (__lane_id() == 0) { atomicAdd(&s_prof[2 * (i)], (unsigned long long)((v1) - (v0))); atomicAdd(&s_prof[2 * (i) + 1], 1ull); } } while (0)
#else
#define PH_T(v) do { } while (0)
#define PH_ADD(v0, v1, i) do { } while (0)
#endif

namespace hpmvs {

// ------------------------------------------------------------------------------------------
// wave64 helpers

__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }

__device__ __forceinline__ float bcast_f(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}
__device__ __forceinline__ int bcast_i(int v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }
__device__ __forceinline__ double bcast_d(double v, int src_lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src_lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src_lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ const uint8_t* bcast_p(const uint8_t* p, int src_lane) {
    const unsigned long long b = (unsigned long long)p;
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(b & 0xffffffffull), src_lane);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(b >> 32), src_lane);
    return (const uint8_t*)(((unsigned long long)hi << 32) | lo);
}

// `(int)x` as the reference's x86 code executes it (cvttss2si / cvttsd2si): truncation towards zero, and the "integer indefinite"
// 0x80000000 for a NaN or a value outside the int range -- where the GPU's v_cvt_i32_* gives 0 / saturates.  Used wherever a
// float that a hostile patch can make non-finite becomes a pyramid level or a pixel index that GATES something (round 6: a NaN scale
// made addImages attach views on the device -- level "0" passes its level gate -- that the reference, with level INT_MIN, skips).
__device__ __forceinline__ int cvt_i32_x86(float v) { return (v >= -2147483648.0f && v < 2147483648.0f) ? (int)v : (int)0x80000000; }
__device__ __forceinline__ int cvt_i32_x86(double v) { return (v > -2147483649.0 && v < 2147483648.0) ? (int)v : (int)0x80000000; }

// ------------------------------------------------------------------------------------------
// per-lane scalar helpers (float32 in the reference's evaluation order)

__device__ __forceinline__ float dot3f(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
__device__ __forceinline__ void cross3f(const float* a, const float* b, float* r) {
    const float r0 = a[1] * b[2] - a[2] * b[1];
    const float r1 = a[2] * b[0] - a[0] * b[2];
    const float r2 = a[0] * b[1] - a[1] * b[0];
    r[0] = r0; r[1] = r1; r[2] = r2;
}
__device__ __forceinline__ void normalized3f(const float* a, float* r) {
    const float n2 = dot3f(a, a);
    if (n2 > 0.0f) { const float n = sqrtf(n2); r[0] = a[0] / n; r[1] = a[1] / n; r[2] = a[2] / n; }
    else { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
}

// PatchOptimizer::calculatePatchAxis (PatchOptimizer.cpp:532-548)
__device__ __forceinline__ void calc_patch_axes(const float* cam_xaxis, const float* cam_yaxis, const float* n,
                                                float scale, float* px, float* py, float* pz) {
    float t[3], x[3], y[3];
    normalized3f(n, pz);
    cross3f(pz, cam_xaxis, t); normalized3f(t, y);
    cross3f(y, pz, t); normalized3f(t, x);
    for (int k = 0; k < 3; k++) { x[k] *= scale; y[k] *= scale; }
    float yn[3], yan[3];
    normalized3f(y, yn); normalized3f(cam_yaxis, yan);
    const float fs = dot3f(yn, yan);
    for (int k = 0; k < 3; k++) { px[k] = x[k]; py[k] = y[k] * fs; }
}

// Camera::project at level 0 (Camera.h:45-62), X.w = 1
__device__ __forceinline__ void project0(const float* P, float X0, float X1, float X2, float& u, float& v) {
    const float r0 = ((P[0] * X0 + P[1] * X1) + P[2] * X2) + P[3];
    const float r1 = ((P[4] * X0 + P[5] * X1) + P[6] * X2) + P[7];
    const float r2 = ((P[8] * X0 + P[9] * X1) + P[10] * X2) + P[11];
    const float lo = -2147483648.0f, hi = 2147483648.0f;  // (float)(INT_MIN+3.0f), (float)(INT_MAX-3.0f)
    const float uu = fmaxf(lo, fminf(hi, r0 / r2)), vv = fmaxf(lo, fminf(hi, r1 / r2));
    const bool behind = r2 <= 0.0f;
    u = behind ? -65535.0f : uu;
    v = behind ? -65535.0f : vv;
}

// Camera::getLeveli (Camera.cpp:92-99): clamp(round((float)log2(r)), 0, max_level) with r = (double)(scale*fsum) / (2.0*fz),
// evaluated as a count of thresholds r >= lvl_thr[k] (DevScene: the smallest double whose logarithm, rounded to float as
// the reference's `float getLevel()` rounds it, reaches k + 0.5): no logarithm on the device, and the same level as the
// reference also for the r a hair below 2^(k+1/2) whose float-rounded log2 is already k + 0.5 (rounds 1-3
// compared r with 2^(k+1/2) itself and gave k there; tests/test_gpu_photometric.py walks scales across those bands).
__device__ __forceinline__ int level_clamped(const DevScene& sc, float scale, float fsum, float fz, int max_level) {
    const double r = (double)(scale * fsum) / (2.0 * (double)fz);
    int lvl = 0;
#pragma unroll
    for (int k = 0; k < kMaxLevels - 1; k++)
        if (k < max_level && r >= sc.lvl_thr[k]) lvl++;
    return lvl;
}

struct Prep {
    int ok;                // gates passed
    float tlx, tly;        // top-left sample position (level-l pixels)
    float dxx, dxy;        // step per xx
    float dyx, dyy;        // step per yy
    const uint8_t* pix;    // level-l image
    int W;                 // level-l width
};

// The per-image, patch-uniform part of sampleTexture (PatchOptimizer.cpp:476-508): angle gate, pyramid level, projected
// window, bounding-box gate -- in two steps, so that the evaluator can have the loads of ALL its job sets in flight
// before it computes any of them: load_view issues the view's table loads, prep_from_view does the arithmetic
// (branch-free: the gates only decide `ok` at the end, exactly as sampleTexture's early returns do;
//  level sizes as w0 >> l: the pyramid halves with integer division, Image.cpp:55-63).  The pointer of the chosen
// level comes from the table too: a view's levels are one allocation and their offsets travel with the view's row.
constexpr int kGrabLevels = 5;  // sampleTexture clamps its level to MAXLEVEL - 1 <= 4 (HpmvsOptions.h: MAXLEVEL = 5); deeper options take the load
struct ViewRegs { float P[12]; float cx, cy, cz, fsum; int w0, h0; const uint8_t* pix0; const DevView* vp; };
__device__ __forceinline__ ViewRegs load_view(const DevView& V) {
    ViewRegs r;
#pragma unroll
    for (int k = 0; k < 12; k++) r.P[k] = V.P0[k];
    r.cx = V.center[0]; r.cy = V.center[1]; r.cz = V.center[2];
    r.fsum = V.fsum; r.w0 = V.w[0]; r.h0 = V.h[0];
    r.pix0 = V.pix[0];
    r.vp = &V;
    return r;
}
__device__ __forceinline__ Prep prep_from_view(const DevScene& sc, const ViewRegs& V, const DevOptions& o, const float* c, float scale,
                                               const float* pxa, const float* pya, const float* zax) {
    Prep r;
    const float d0 = V.cx - c[0], d1 = V.cy - c[1], d2 = V.cz - c[2];
    const float n2 = (d0 * d0 + d1 * d1) + d2 * d2;  // w components cancel (1 - 1)
    const float nn = sqrtf(n2);
    const float g_unit = ((d0 / nn) * zax[0] + (d1 / nn) * zax[1]) + (d2 / nn) * zax[2];
    const float g_raw = (d0 * zax[0] + d1 * zax[1]) + d2 * zax[2];
    const float g = (n2 > 0.0f) ? g_unit : g_raw;
    const bool angle_ok = !((double)g < o.cos_max_d);
    const int lvl = level_clamped(sc, scale, V.fsum, nn, o.MAXLEVEL - 1);
    {   // the chosen level's pixels: pix[0] + off[lvl].  A view's levels are one allocation, level k + 1 behind level k
        // at the next multiple of 256 bytes past its 3 * w * h + 16 bytes (capi.hip, scene_set_view): the offset is
        // arithmetic on the level-0 size, so no load depends on the level
        unsigned ofs = 0u;
#pragma unroll
        for (int k = 0; k < kGrabLevels - 1; k++) {
            const unsigned nb = (3u * (unsigned)(V.w0 >> k) * (unsigned)(V.h0 >> k) + 16u + 255u) & ~255u;
            ofs += (k < lvl) ? nb : 0u;
        }
        r.pix = (lvl < kGrabLevels) ? V.pix0 + ofs : V.vp->pix[lvl];
    }
    float cu, cv, xu, xv, yu, yv;
    project0(V.P, c[0], c[1], c[2], cu, cv);
    project0(V.P, c[0] + pxa[0], c[1] + pxa[1], c[2] + pxa[2], xu, xv);
    project0(V.P, c[0] + pya[0], c[1] + pya[1], c[2] + pya[2], yu, yv);
    const float s = __int_as_float((127 - lvl) << 23);  // 2^-lvl, exact
    cu *= s; cv *= s;
    const float dxx = xu * s - cu, dxy = xv * s - cv;
    const float dyx = yu * s - cu, dyy = yv * s - cv;
    const float hs = 3.5f;
    const float tlx = (cu - hs * dxx) - hs * dyx, tly = (cv - hs * dxy) - hs * dyy;
    const float trx = (cu + hs * dxx) - hs * dyx, try_ = (cv + hs * dxy) - hs * dyy;
    const float blx = (cu - hs * dxx) + hs * dyx, bly = (cv - hs * dxy) + hs * dyy;
    const float brx = (cu + hs * dxx) + hs * dyx, bry = (cv + hs * dxy) + hs * dyy;
    const float mnx = fminf(fminf(fminf(tlx, trx), blx), brx), mny = fminf(fminf(fminf(tly, try_), bly), bry);
    const float mxx = fmaxf(fmaxf(fmaxf(tlx, trx), blx), brx), mxy = fmaxf(fmaxf(fmaxf(tly, try_), bly), bry);
    const int W = V.w0 >> lvl, H = V.h0 >> lvl;
    const float m = 3.0f;
    const bool box_ok = !(mnx < m || mny < m || mxx >= (float)(W - 3) || mxy >= (float)(H - 3));
    r.ok = (angle_ok && box_ok) ? 1 : 0;
    r.tlx = tlx; r.tly = tly; r.dxx = dxx; r.dxy = dxy; r.dyx = dyx; r.dyy = dyy;
    r.W = W;
    return r;
}

// 8 unaligned bytes from global memory (address space 1, so the compiler emits
// global_load_dwordx2 rather than a flat load for a pointer that travelled through readlane)
__device__ __forceinline__ uint2 load8(const uint8_t* p) {
    typedef __attribute__((address_space(1))) const uint8_t gbyte;
    gbyte* g = (gbyte*)(unsigned long long)p;
    uint2 v;
    __builtin_memcpy(&v, g, 8);
    return v;
}

// Correctly rounded float32 division through float64.  For float a, w the quotient a/w lies at a relative
// distance > 2^-49 from every rounding boundary of the float32 format (a 25-bit midpoint m would need
// a = m*w, i.e. |a - m*w| >= 1 unit of a 49-bit product), while (double)a * RN(1/(double)w) is within
// 2^-52 of a/w: rounding it to float32 therefore gives exactly RN(a/w), the value the CPU's divss
// returns -- with one reciprocal per DIVISOR instead of a ~11-instruction IEEE division sequence per
// quotient (the divisor is shared by 147 texels, or is a constant).  Results stay far from the
// denormal range here (|a| is 0 or >= 2^-20, w <= 256).
__device__ __forceinline__ double recip64(float w) { return 1.0 / (double)w; }
__device__ __forceinline__ float div_by(float a, double inv_w) { return (float)((double)a * inv_w); }

__device__ __forceinline__ float robustincc(float rhs) { return rhs / (1 + 3 * rhs); }  // PatchOptimizer.h:92-94

// ------------------------------------------------------------------------------------------
// ROW-LANE exact-order evaluator: every photometric request a wave has pending, 9 grabs per pass, no LDS.
//
// BOBYQA on this objective amplifies a 1e-9 relative perturbation of f into a different basin for 7-8 % of
// patches (tests/sensitivity_probe.py), so the objective has to be the reference's float32 arithmetic bit for
// bit, including the SEQUENTIAL summation order of Patch2d::normalize / dot (Patch2d.hpp:37-84) and the
// incremental sample stepping of sampleTexture (PatchOptimizer.cpp:511-526).  A wave reduction cannot do
// that.  Rounds 1-2 gave a grab 49 lanes (one per texel) and handed the sequential sums to single lanes through
// an LDS staging buffer: 5 barriers, ~280 issued instructions per grab, 8 of 64 lanes busy during the sums, and
// 5.4 KB of LDS per wavefront that the optimiser state could not use.  Here a grab gets SEVEN lanes, one per
// ROW of the 7x7 window (lane = 7*group + yy, nine groups per wavefront):
//   * lane yy steps to its row start with the reference's incremental adds and samples its 7 texels itself
//     (14 independent 8-byte tap loads in flight per lane, all of them before the first conversion);
//   * a sequential sum over the window in pixel order is a chain through the 7 lanes of the group: lane yy adds
//     its 7 (or 21) terms in order onto the value lane yy-1 hands over (v_add_f32_dpp wave_shr:1: the hand-over
//     is fused into the first add); seven stages, every lane executes every stage, the stage's own lane keeps
//     the result.  Three channel means run as three independent chains (ILP), then one chain for the squared
//     deviations and one 147-term chain for the dot product -- exactly Patch2d.hpp's operation order;
//   * a group serves ONE request from start to end: the reference image first (its normalised texels stay in
//     the group's registers), then the other usable images in ascending index, so the per-request
//     accumulations (objective_fn's float64 sum, setINCCs' values) happen in the reference's loop order inside
//     the group and no texture ever crosses lanes.  Idle groups pull the next pending request.
// Per grab ~120 issued instructions, no LDS, no barrier; the optimiser state gets the whole 160 KB.
//
// Before the passes every (request, image) pair is a "job" of its own lane (up to two sets of 64 jobs): the
// job lane runs prep_from_view (gates, level, projected window) with the request's data fetched from the
// requesting lane by ds_bpermute; the groups fetch the geometry of their current grab from the job lane.
struct PackedReq {   // registers of the requesting lane (valid where its `pending` bit is set)
    int obj_mode;    // 1: objective_fn (reference image = list[0]); 0: setINCCs against list[ref]
    int ref, robust, nimg;
    float thr;       // filterImagesNCC threshold (INCC requests)
    float c[3], zax[3], scale, px[3], py[3];  // zax: patch z axis (OBJ) or the raw normal (INCC), as sampleTexture's angle gate uses them
};
struct PackedRes {
    double obj;               // OBJ
    unsigned long long keep;  // INCC: bit v set <=> inccs[v] < 1 - thr
    float incc_sum;           // INCC: sequential float sum of inccs
    int ngrabs;               // added to
};

__device__ __forceinline__ void wave_lds_sync() {
    // single wavefront: DS operations execute in issue order; this only stops the compiler from
    // moving LDS accesses across the hand-over between lanes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float shfl_f(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}
__device__ __forceinline__ int shfl_i(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
__device__ __forceinline__ unsigned long long bcast_u64(unsigned long long v, int src_lane) {
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(v & 0xffffffffull), src_lane);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(v >> 32), src_lane);
    return ((unsigned long long)hi << 32) | lo;
}

struct Taps { uint2 a, b; float f00, f01, f10, f11; };

// Image::getColor (Image.h:104-113) in its evaluation order
__device__ __forceinline__ void texel(const Taps& t, float& r, float& g, float& b) {
    // (Image::getColor starts from 0 and adds: the first sum is a sum of products of non-negative values, never -0,
    //  so 0 + it is it)
    r = (float)(t.a.x & 0xff) * t.f00 + (float)(t.b.x & 0xff) * t.f01;
    g = (float)((t.a.x >> 8) & 0xff) * t.f00 + (float)((t.b.x >> 8) & 0xff) * t.f01;
    b = (float)((t.a.x >> 16) & 0xff) * t.f00 + (float)((t.b.x >> 16) & 0xff) * t.f01;
    r += (float)(t.a.x >> 24) * t.f10 + (float)(t.b.x >> 24) * t.f11;
    g += (float)(t.a.y & 0xff) * t.f10 + (float)(t.b.y & 0xff) * t.f11;
    b += (float)((t.a.y >> 8) & 0xff) * t.f10 + (float)((t.b.y >> 8) & 0xff) * t.f11;
}

// texel() for two texels A, B in the halves of float2 values.  (The `0 +` that texel() starts each channel with is
// left out here and there: the terms are products of non-negative values, never -0, so adding them to +0 changes nothing.)
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void texel_pair(uint2 aA, uint2 bA, float wxA, float wyA, uint2 aB, uint2 bB, float wxB, float wyB,
                                           float* outA, float* outB) {
    const v2f dx1 = {wxA, wxB}, dy1 = {wyA, wyB};
    const v2f one = {1.0f, 1.0f};
    const v2f dx0 = one - dx1, dy0 = one - dy1;
    const v2f f00 = dx0 * dy0, f01 = dx0 * dy1, f10 = dx1 * dy0, f11 = dx1 * dy1;
#define HPMVS_B(v, sh) (float)(((v) >> (sh)) & 0xffu)
    const v2f r00 = {HPMVS_B(aA.x, 0), HPMVS_B(aB.x, 0)}, r01 = {HPMVS_B(bA.x, 0), HPMVS_B(bB.x, 0)};
    const v2f g00 = {HPMVS_B(aA.x, 8), HPMVS_B(aB.x, 8)}, g01 = {HPMVS_B(bA.x, 8), HPMVS_B(bB.x, 8)};
    const v2f b00 = {HPMVS_B(aA.x, 16), HPMVS_B(aB.x, 16)}, b01 = {HPMVS_B(bA.x, 16), HPMVS_B(bB.x, 16)};
    const v2f r10 = {(float)(aA.x >> 24), (float)(aB.x >> 24)}, r11 = {(float)(bA.x >> 24), (float)(bB.x >> 24)};
    const v2f g10 = {HPMVS_B(aA.y, 0), HPMVS_B(aB.y, 0)}, g11 = {HPMVS_B(bA.y, 0), HPMVS_B(bB.y, 0)};
    const v2f b10 = {HPMVS_B(aA.y, 8), HPMVS_B(aB.y, 8)}, b11 = {HPMVS_B(bA.y, 8), HPMVS_B(bB.y, 8)};
#undef HPMVS_B
    const v2f r = (r00 * f00 + r01 * f01) + (r10 * f10 + r11 * f11);
    const v2f g = (g00 * f00 + g01 * f01) + (g10 * f10 + g11 * f11);
    const v2f b = (b00 * f00 + b01 * f01) + (b10 * f10 + b11 * f11);
    outA[0] = r.x; outA[1] = g.x; outA[2] = b.x;
    outB[0] = r.y; outB[1] = g.y; outB[2] = b.y;
}

constexpr int kGroups = 9;               // grabs per pass: nine 7-lane groups, lane 63 is spare
#ifndef HPMVS_JOB_SETS
#define HPMVS_JOB_SETS 2   // measured at C2: 2 sets 297 ms, 3 sets 319 ms, 4 sets 309 ms (fewer passes, but the per-pass fetch and
                           // the registers of the extra sets cost more than the passes they save)
#endif
constexpr int kJobSets = HPMVS_JOB_SETS;  // job lanes are used this many times over: up to 64 * kJobSets (request, image) pairs per batch
constexpr int kRowVals = 3 * kTex;       // 21 float32 values (7 texels x RGB) per lane
constexpr int kFinChunk = 8;             // images per step of the requesting lanes' accumulation (most requests have <= 8)

// value held by lane - 1 (lane 0 receives 0): DPP wave_shr:1, folded by the compiler into the consuming add
__device__ __forceinline__ float lane_prev(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}

// Sequential float32 sum over a group's window in pixel order: lane yy contributes v[0..N-1] in order, lane
// yy+1 continues from lane yy's total.  ((0 + row0.v0) + row0.v1 ...) + row1.v0 ... -- the summation order of
// Patch2d::normalize / dot.  The total is valid on lane yy == 6 of the group.
// (No select per stage: after stage s the value is right on the lanes with yy == s -- by induction, since such a
//  lane read its left neighbour's value right after stage s - 1 -- and whatever the other lanes hold is never read
//  when it matters: a lane's value is consumed by its right neighbour in the very next stage, and the final stage
//  leaves the total on yy == 6.)
template <int N>
__device__ __forceinline__ float row_chain(const float (&v)[N]) {
    float out = 0.0f;
#pragma unroll
    for (int s = 0; s < kTex; s++) {
        float a = (s == 0) ? 0.0f : lane_prev(out);
#pragma unroll
        for (int i = 0; i < N; i++) a += v[i];
        out = a;
    }
    return out;
}

// two such chains at once, in the halves of a float2 (same additions per half, half the add instructions; the
// hand-over is two DPP moves instead of being folded into the first add)
template <int N>
__device__ __forceinline__ void row_chain_pair(const float (&va)[N], const float (&vb)[N], float& outa, float& outb) {
    v2f out = {0.0f, 0.0f};
#pragma unroll
    for (int s = 0; s < kTex; s++) {
        v2f a;
        if (s == 0) { a.x = 0.0f; a.y = 0.0f; }
        else { a.x = lane_prev(out.x); a.y = lane_prev(out.y); }
#pragma unroll
        for (int i = 0; i < N; i++) { const v2f t = {va[i], vb[i]}; a = a + t; }
        out = a;
    }
    outa = out.x; outb = out.y;
}

// The three window sums of a grab.  Parity mode (the default, what every test runs): Patch2d's sequential order through
// the lane chains above.  HPMVS_FAST_SUMS (an opt-in THROUGHPUT build of the kernel, `HPMVS_SUMS=tree`): every lane sums
// its own row first and the seven row sums are added in row order -- 7 (or 21) + 6 adds where the chains need 49 (147),
// the same mathematical sum with a different rounding order, i.e. one more member of the family of orders a real
// hpmvs binary may have (Eigen's reductions already differ between ISAs).  Not bit-identical to the
// oracle; profiles/r04/sums_modes.json holds what it costs in agreement and buys in time.
#ifndef HPMVS_FAST_SUMS
#define HPMVS_FAST_SUMS 0
#endif
#ifndef HPMVS_SPLIT_PREP
#define HPMVS_SPLIT_PREP 1
#endif
#ifndef HPMVS_SINGLE_FAST
#define HPMVS_SINGLE_FAST 1
#endif
#ifndef HPMVS_TAKE_OVER
#define HPMVS_TAKE_OVER 1  // idle groups take over images of the busiest ones (wave_eval_rows, step 1b)
#endif
template <int N>
__device__ __forceinline__ float window_sum(const float (&v)[N]) {
#if HPMVS_FAST_SUMS
    float a = v[0];
#pragma unroll
    for (int i = 1; i < N; i++) a += v[i];
    float out = a;
#pragma unroll
    for (int s = 1; s < kTex; s++) out = lane_prev(out) + a;  // after step s the lanes with yy >= s hold rows yy-s .. yy
    return out;
#else
    return row_chain<N>(v);
#endif
}
__device__ __forceinline__ void mean_chains(const float (&vr)[kTex], const float (&vg)[kTex], const float (&vb)[kTex],
                                            float& s0, float& s1, float& s2) {
#if HPMVS_FAST_SUMS
    s0 = window_sum<kTex>(vr); s1 = window_sum<kTex>(vg); s2 = window_sum<kTex>(vb);
#else
    // the red and green chains run in the two halves of one register pair (v_pk_add_f32), the blue one beside them
    row_chain_pair<kTex>(vr, vg, s0, s1);
    s2 = row_chain<kTex>(vb);
#endif
}

#ifndef HPMVS_SCAN_BATCHES
#define HPMVS_SCAN_BATCHES 1
#endif
// inclusive prefix sum over the 64 lanes (DPP: four shifts inside the rows of 16 lanes, then the rows' totals)
__device__ __forceinline__ int wave_scan_add(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2 and 3
    return v;
}

// maximum over the lanes of `v`
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
    return v;
}

// Where the per-image results of a batch's requests wait until the requesting lanes add them up in the reference's
// order: one row of floats per requesting lane (row of lane p = base + p * stride).  Global memory in the refinement
// kernel (the launch workspace, L2 resident: a CU's own stores and loads are coherent through its vector cache) and in
// the stand-alone setINCCs kernel (its output rows); LDS in the stand-alone objective kernel.
struct GlobalRows {
    float* base; int stride;
    typedef __attribute__((address_space(1))) float gfloat;
    __device__ __forceinline__ void store(int p, int k, float v) const { ((gfloat*)(unsigned long long)(base + (size_t)p * stride))[k] = v; }
    __device__ __forceinline__ float load(int p, int k) const { return ((const gfloat*)(unsigned long long)(base + (size_t)p * stride))[k]; }
};
struct LdsRows {
    float* base; int stride;
    typedef __attribute__((address_space(3))) float lfloat;
    __device__ __forceinline__ void store(int p, int k, float v) const { ((lfloat*)base)[p * stride + k] = v; }
    __device__ __forceinline__ float load(int p, int k) const { return ((const lfloat*)base)[p * stride + k]; }
};

// list: the calling lane's own image list (ids; read only for lanes with a pending request).  Must be called by
// all 64 lanes.  rows: see above (the row of a requesting lane holds at least min(its image count, stride) floats).
// WRITE_INCCS (the stand-alone setINCCs kernel): the rows ARE the output -- the values of the usable images and the 0 of
// the reference image stay there, and out.keep returns the mask of the grabbed images instead of the thresholded one.
//
// Round 4: a grab's result is no longer accumulated inside the group that computed it.  The group's first lane stores it
// (1 - ncc, or its robust form) at [request][image] of `rows`, and when the batch's passes are over every requesting lane
// adds up its own row in index order -- objective_fn's float64 sum (PatchOptimizer.cpp:299-309), setINCCs' values and
// std::accumulate over them (:462-472, :161-167) -- lane-parallel over the batch's requests.  Since the ORDER of the
// reference's accumulations now lives in that last step only, any group may compute any image of any request whose
// normalised reference texels it holds: a group that runs out of work takes the upper half of the images the busiest
// group still has to do (with a copy of its reference texels: 21 ds_bpermute), instead of idling until the batch's
// longest request is through.  At BASELINE configs[2] 12.5 % of the group-passes were idle that way; for a lone patch
// (the ramp-down of a launch, single-patch callers) the images of its one request now run side by side.
template <bool WRITE_INCCS, typename ListT, typename Rows>
__device__ __forceinline__ void wave_eval_rows(const DevScene& sc, const DevOptions& o, const ListT* list,
                                               unsigned long long pending, const PackedReq& q, PackedRes& out,
                                               const Rows rows) {
    const int lane = lane_id();
    const int grp = lane / kTex;            // 0..9; 9 = the spare lane
    const int yy = lane - grp * kTex;       // window row of this lane
    const bool glane = grp < kGroups;
    const int last = glane ? grp * kTex + (kTex - 1) : lane;  // lane that ends the group's chains
    const int gl0 = lane - yy;              // first lane of my group
    const int q_refsel = q.obj_mode ? 0 : q.ref;
    const int q_userob = (q.obj_mode || q.robust) ? 1 : 0;
    const unsigned long long listbits = (unsigned long long)list;
    const int list_lo = (int)(listbits & 0xffffffffull), list_hi = (int)(listbits >> 32);

    unsigned long long rem = pending;
    while (rem) {
        // ---- a batch of whole requests whose jobs fit two sets of 64 job lanes (at least one request)
        // When the pending requests need several batches they are cut EVENLY: a batch's pass count is about
        // max(longest request, grabs / 9) + 1, so a last batch with a handful of requests wastes most of its groups.
#if HPMVS_SCAN_BATCHES
        // (round 5) the batch, its jobs and every request's first job from ONE prefix sum over the pending requests' image
        // counts (DPP scan) instead of three scalar loops over the requests (a v_readlane and a dozen scalar instructions
        // per request and loop: ~4 k cycles per batch with 50 pending requests).  The same batches as the loops form.
        const bool inrem = (rem >> lane) & 1ull;
        const int mynp = inrem ? q.nimg : 0;
        const int incl = wave_scan_add(mynp);
        const int excl = incl - mynp;                      // jobs of the pending requests before mine
        const int j_rem = bcast_i(incl, 63);
        const int n_batches = (j_rem + 64 * kJobSets - 1) / (64 * kJobSets);
        const int j_target = n_batches > 1 ? (j_rem + n_batches - 1) / n_batches : 64 * kJobSets;
        const int first = __ffsll((long long)rem) - 1;
        // the loop's rule: requests are added in lane order until one would overflow the job lanes or the target is reached
        const unsigned long long stop = __ballot(inrem && lane != first && (incl > 64 * kJobSets || excl >= j_target));
        const unsigned long long batch = stop ? (rem & ((1ull << (__ffsll((long long)stop) - 1)) - 1ull)) : rem;
        const int lastb = 63 - __clzll(batch);
        const int J = bcast_i(incl, lastb);
        rem &= ~batch;
        PH_T(t1);
        // ---- jobs of this lane: job 64 * k + lane of set k -> (request, image index)
        const int myoff = excl;  // requesting lanes: index of my request's first job
        int jp[kJobSets], ji[kJobSets];
        {
            // job j belongs to the request p with excl[p] <= j < incl[p]: the smallest lane whose inclusive sum exceeds j
            // (incl is non-decreasing over the lanes: lanes without a pending request add 0) -- a binary search, six
            // ds_bpermute per job set, the sets' searches interleaved
            int lo[kJobSets];
#pragma unroll
            for (int k = 0; k < kJobSets; k++) lo[k] = 0;
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) {
                int v[kJobSets];
#pragma unroll
                for (int k = 0; k < kJobSets; k++) v[k] = shfl_i(incl, lo[k] + step - 1);
#pragma unroll
                for (int k = 0; k < kJobSets; k++) lo[k] += (v[k] <= lane + 64 * k) ? step : 0;
            }
#pragma unroll
            for (int k = 0; k < kJobSets; k++) {
                const int j = lane + 64 * k;
                const int e = shfl_i(excl, lo[k] & 63);
                const bool has = j < J;
                jp[k] = has ? lo[k] : -1;
                ji[k] = has ? j - e : 0;
            }
        }
#else
        int j_rem = 0;
        for (unsigned long long t = rem; t; t &= t - 1ull) j_rem += bcast_i(q.nimg, __ffsll((long long)t) - 1);
        const int n_batches = (j_rem + 64 * kJobSets - 1) / (64 * kJobSets);
        const int j_target = n_batches > 1 ? (j_rem + n_batches - 1) / n_batches : 64 * kJobSets;
        unsigned long long batch = 0ull;
        int J = 0;
        for (unsigned long long t = rem; t; t &= t - 1ull) {
            const int p = __ffsll((long long)t) - 1;
            const int np = bcast_i(q.nimg, p);
            if (batch && (J + np > 64 * kJobSets || J >= j_target)) break;
            batch |= 1ull << p;
            J += np;
        }
        rem &= ~batch;
        PH_T(t1);
        // ---- jobs of this lane: job 64 * k + lane of set k -> (request, image index)
        int myoff = 0;  // requesting lanes: index of my request's first job
        int jp[kJobSets], ji[kJobSets];
#pragma unroll
        for (int k = 0; k < kJobSets; k++) { jp[k] = -1; ji[k] = 0; }
        {
            int off = 0;
            for (unsigned long long t = batch; t; t &= t - 1ull) {
                const int p = __ffsll((long long)t) - 1;
                const int np = bcast_i(q.nimg, p);
                if (lane == p) myoff = off;
#pragma unroll
                for (int k = 0; k < kJobSets; k++) {
                    const int j = lane + 64 * k;
                    if (j >= off && j < off + np) { jp[k] = p; ji[k] = j - off; }
                }
                off += np;
            }
        }
#endif
        // Staged over all job sets in use, so that the loads of one set hide behind those of the others: request data
        // from the requesting lanes (ds_bpermute), image ids, the views' tables, then the arithmetic.
        Prep PS[kJobSets];
        unsigned long long okS[kJobSets];
        int pix_lo[kJobSets], pix_hi[kJobSets];
        {
            float jc[kJobSets][3], jz[kJobSets][3], jx[kJobSets][3], jy[kJobSets][3], jscale[kJobSets];
            int jid[kJobSets];
#pragma unroll
            for (int k = 0; k < kJobSets; k++) {
                jid[k] = 0; jscale[k] = 1.0f;
                for (int a3 = 0; a3 < 3; a3++) { jc[k][a3] = 0.0f; jz[k][a3] = 0.0f; jx[k][a3] = 0.0f; jy[k][a3] = 0.0f; }
                if (J > 64 * k) {  // (wave-uniform)
                    const int src = jp[k] >= 0 ? jp[k] : 0;
#pragma unroll
                    for (int a3 = 0; a3 < 3; a3++) {
                        jc[k][a3] = shfl_f(q.c[a3], src); jz[k][a3] = shfl_f(q.zax[a3], src);
                        jx[k][a3] = shfl_f(q.px[a3], src); jy[k][a3] = shfl_f(q.py[a3], src);
                    }
                    jscale[k] = shfl_f(q.scale, src);
                    const unsigned long long lb = ((unsigned long long)(unsigned int)shfl_i(list_hi, src) << 32) |
                                                  (unsigned long long)(unsigned int)shfl_i(list_lo, src);
                    // (a lane without a job prepares view 0 and drops the result: no branch around the arithmetic)
                    if (jp[k] >= 0) jid[k] = (int)((const ListT*)lb)[ji[k]];
                }
            }
            ViewRegs VR[kJobSets];
#pragma unroll
            for (int k = 0; k < kJobSets; k++) {
                VR[k] = ViewRegs{};
                if (k == 0 || J > 64) VR[k] = load_view(sc.views[jid[k]]);
            }
            // (wave-uniform branch: a lone request, or the last few of a round, fill the first set only; when the others are in
            //  use the sets' arithmetic stays one straight block, so that the compiler interleaves their dependent chains)
#if HPMVS_SPLIT_PREP == 2
#pragma unroll
            for (int k = 0; k < kJobSets; k++) {
                PS[k] = Prep{};
                if (k == 0 || J > 64) PS[k] = prep_from_view(sc, VR[k], o, jc[k], jscale[k], jx[k], jy[k], jz[k]);
            }
#else
            if (kJobSets == 1 || !HPMVS_SPLIT_PREP || J > 64) {
#pragma unroll
                for (int k = 0; k < kJobSets; k++) PS[k] = prep_from_view(sc, VR[k], o, jc[k], jscale[k], jx[k], jy[k], jz[k]);
            } else {
                PS[0] = prep_from_view(sc, VR[0], o, jc[0], jscale[0], jx[0], jy[0], jz[0]);
#pragma unroll
                for (int k = 1; k < kJobSets; k++) PS[k] = Prep{};
            }
#endif
#pragma unroll
            for (int k = 0; k < kJobSets; k++) {
                PS[k].ok = (J > 64 * k && jp[k] >= 0) ? PS[k].ok : 0;
                okS[k] = __ballot(PS[k].ok != 0);
                const unsigned long long pb = (unsigned long long)PS[k].pix;
                pix_lo[k] = (int)(pb & 0xffffffffull); pix_hi[k] = (int)(pb >> 32);
            }
        }

        // ---- per request, on its requesting lane: which images are usable
        const bool inbatch = (batch >> lane) & 1ull;
        unsigned long long myok = 0ull;
        if (inbatch) {
            // bits myoff .. myoff + nimg - 1 of the concatenated ok masks of the sets
            unsigned long long lo = 0ull;
            const int w = myoff >> 6, sh = myoff & 63;
#pragma unroll
            for (int k = 0; k < kJobSets; k++) {
                if (k == w) lo |= okS[k] >> sh;
                if (k == w + 1 && sh > 0) lo |= okS[k] << (64 - sh);
            }
            myok = (q.nimg >= 64) ? lo : (lo & ((1ull << q.nimg) - 1ull));
        }
        const bool refok = inbatch && q.nimg > 0 && ((myok >> q_refsel) & 1ull);
        unsigned long long todo = __ballot(refok);
        const int mylen = __popcll(myok);  // grabs of my request
        const unsigned long long len6 = __ballot(mylen >= 6), len4 = __ballot(mylen >= 4);
#if defined(HPMVS_PROFILE_PHASES) && defined(__HIP_DEVICE_COMPILE__)
        {   // pass-count statistics of this batch: longest request, ideal pass count, long non-OBJ requests
            int mx = 0, tot = 0, longreq = 0;
            for (unsigned long long t = todo; t; t &= t - 1ull) {
                const int pp = __ffsll((long long)t) - 1;
                const int l = bcast_i(mylen, pp);
                mx = l > mx ? l : mx; tot += l;
                longreq += (l > 8 && !bcast_i(q.obj_mode, pp)) ? 1 : 0;
            }
            if (lane == 0) {
                atomicAdd(&s_prof[60], (unsigned long long)mx); atomicAdd(&s_prof[61], (unsigned long long)((tot + kGroups - 1) / kGroups));
                atomicAdd(&s_prof[62], 1ull); atomicAdd(&s_prof[63], (unsigned long long)longreq);
            }
        }
#endif
        PH_T(t2);
        PH_ADD(t1, t2, 13);

        // ---- the grab my group is on (identical on the 7 lanes of a group)
        int g_req = -1, g_off = 0, g_ref = 0, g_userob = 0, g_cur = 0;
        bool g_isref = false;
        unsigned long long g_ok = 0ull;  // images of the request this group still has to do after the current one
        float refn[kRowVals];  // my row of the normalised reference texture
#pragma unroll
        for (int i = 0; i < kRowVals; i++) refn[i] = 0.0f;

        bool single = HPMVS_SINGLE_FAST && __popcll(todo) == 1;  // (wave-uniform)
        for (;;) {
            PH_T(t3);
            int rsrc = lane;     // lane that holds (after step 4 of this pass at the latest) my row of the reference texels I correlate with
            bool moved = false;  // (wave-uniform) groups took images over in this pass: their reference texels travel before step 5
            // ---- 1. idle groups take the next pending requests, long requests first (fewest idle group-passes at the
            //         end of the batch): three length classes
            {
                unsigned long long idle = __ballot(glane && yy == 0 && g_req < 0);
                while (todo && idle) {
                    const unsigned long long c6 = todo & len6, c4 = todo & len4;
                    const int p = __ffsll((long long)(c6 ? c6 : (c4 ? c4 : todo))) - 1;
                    todo &= ~(1ull << p);
                    const int gl = __ffsll((long long)idle) - 1;  // first lane of the group
                    idle &= idle - 1ull;
                    const int r_off = bcast_i(myoff, p), r_ref = bcast_i(q_refsel, p), r_rob = bcast_i(q_userob, p);
                    const unsigned long long r_ok = bcast_u64(myok, p);
                    if (gl0 == gl) {
                        g_req = p; g_off = r_off; g_ref = r_ref; g_userob = r_rob;
                        g_ok = r_ok & ~(1ull << r_ref);
                        g_cur = r_ref; g_isref = true;
                    }
                }
                // ---- 1a. the batch is ONE request (a lone patch: the ramp-down of a launch, a single-patch caller): its
                //          reference image stays with the first group, the next eight usable images go to one group each --
                //          what the take-over rounds below arrive at, without their searches
                if (single) {  // (wave-uniform; first pass of the batch only)
                    single = false;
                    const unsigned long long nonref = bcast_u64(g_ok, 0);
                    const int n_other = __popcll(nonref);
                    const int nh = n_other < kGroups - 1 ? n_other : kGroups - 1;
                    if (nh > 0) {
                        const int r_req = bcast_i(g_req, 0), r_off = bcast_i(g_off, 0), r_ref = bcast_i(g_ref, 0), r_rob = bcast_i(g_userob, 0);
                        unsigned long long mine = nonref, rest = nonref;
#pragma unroll
                        for (int i = 1; i < kGroups; i++) {
                            if (i < grp) mine &= mine - 1ull;   // group g: the g-th lowest of the other images
                            if (i <= nh) rest &= rest - 1ull;   // what stays with the first group
                        }
                        if (grp == 0) g_ok = rest;
                        else if (glane && grp <= nh) {
                            g_req = r_req; g_off = r_off; g_ref = r_ref; g_userob = r_rob;
                            g_cur = __ffsll((long long)mine) - 1; g_ok = 0ull; g_isref = false; rsrc = yy;
                        }
                        moved = true;
                        idle = 0ull;
                    }
                }
                // ---- 1b. nothing left to hand out and groups are idle: they take over the upper half of what the
                //          busiest groups still have to do (see the header comment).  The batch ends with the group that
                //          has the most images left, so a round of take-overs pays only if EVERY group at that maximum
                //          gets a helper; the search is scalar code over the nine group leaders (no cross-lane traffic),
                //          the reference texels of all take-overs of a round travel in one set of 21 ds_bpermute.
                while (HPMVS_TAKE_OVER && idle) {  // (only reached with todo == 0)
                    const int left = (glane && g_req >= 0) ? __popcll(g_ok) : 0;   // identical on a group's lanes
                    int best = 0, n_best = 0;
                    unsigned long long vmask = 0ull;  // leaders of the groups at the maximum
#pragma unroll
                    for (int g = 0; g < kGroups; g++) {
                        const int l = bcast_i(left, g * kTex);
                        if (l > best) { best = l; n_best = 0; vmask = 0ull; }
                        if (l == best && l > 0) { n_best++; vmask |= 1ull << (g * kTex); }
                    }
                    if (best < 1 || __popcll(idle) < n_best) break;
                    for (unsigned long long t = vmask; t; t &= t - 1ull) {
                        const int vl = __ffsll((long long)t) - 1;  // the victim's first lane
                        const int tl = __ffsll((long long)idle) - 1;
                        idle &= idle - 1ull;
                        const int take = (best + 1) / 2;
                        unsigned long long hi = bcast_u64(g_ok, vl);
                        for (int i = 0; i < best - take; i++) hi &= hi - 1ull;  // the victim keeps its lowest images
                        const int r_req = bcast_i(g_req, vl), r_off = bcast_i(g_off, vl), r_ref = bcast_i(g_ref, vl), r_rob = bcast_i(g_userob, vl);
                        const int r_src = bcast_i(rsrc, vl);  // where the victim's reference texels are, or will be in this pass (its own lane unless it is a helper of this pass itself)
                        if (gl0 == vl) g_ok &= ~hi;
                        if (gl0 == tl && glane) {
                            g_req = r_req; g_off = r_off; g_ref = r_ref; g_userob = r_rob;
                            g_cur = __ffsll((long long)hi) - 1; g_ok = hi & (hi - 1ull); g_isref = false; rsrc = r_src + yy;
                        }
                    }
                    moved = true;
                }
            }
            const bool act = glane && g_req >= 0;
            if (!__any(act)) break;

            // ---- 2. window geometry of my group's grab, from its job lane
            float tlx, tly, dxx, dxy, dyx, dyy;
            const uint8_t* pix;
            int W;
            {
                const int j = g_off + g_cur;
                const int src = j & 63, set = j >> 6;
                tlx = shfl_f(PS[0].tlx, src); tly = shfl_f(PS[0].tly, src);
                dxx = shfl_f(PS[0].dxx, src); dxy = shfl_f(PS[0].dxy, src);
                dyx = shfl_f(PS[0].dyx, src); dyy = shfl_f(PS[0].dyy, src);
                W = shfl_i(PS[0].W, src);
                int plo = shfl_i(pix_lo[0], src), phi = shfl_i(pix_hi[0], src);
#pragma unroll
                for (int k = 1; k < kJobSets; k++) {
                    if (J > 64 * k) {  // (wave-uniform)
                        const bool sb = set == k;
                        const float b0 = shfl_f(PS[k].tlx, src), b1 = shfl_f(PS[k].tly, src), b2 = shfl_f(PS[k].dxx, src),
                                    b3 = shfl_f(PS[k].dxy, src), b4 = shfl_f(PS[k].dyx, src), b5 = shfl_f(PS[k].dyy, src);
                        const int b6 = shfl_i(PS[k].W, src), b7 = shfl_i(pix_lo[k], src), b8 = shfl_i(pix_hi[k], src);
                        tlx = sb ? b0 : tlx; tly = sb ? b1 : tly; dxx = sb ? b2 : dxx; dxy = sb ? b3 : dxy;
                        dyx = sb ? b4 : dyx; dyy = sb ? b5 : dyy; W = sb ? b6 : W; plo = sb ? b7 : plo; phi = sb ? b8 : phi;
                    }
                }
                pix = (const uint8_t*)(((unsigned long long)(unsigned int)phi << 32) | (unsigned long long)(unsigned int)plo);
            }

            // ---- 3. my row of the window: sampleTexture's stepping (l += dy per row, c += dx per column,
            //         PatchOptimizer.cpp:511-526); all 14 tap loads are issued before the first conversion
            float t[kRowVals];
            {
                // x += step taken `yy` times: fma(m, step, x) with m in {0, 1} is the same single rounding as
                // x + step (m * step is exact) and leaves x untouched for m = 0
                float x = tlx, y = tly;
#pragma unroll
                for (int k = 0; k < kTex - 1; k++) { const float m = (k < yy) ? 1.0f : 0.0f; x = __builtin_fmaf(m, dyx, x); y = __builtin_fmaf(m, dyy, y); }
                uint2 ta[kTex], tb[kTex];
                float wx[kTex], wy[kTex];
                int toff[kTex];
#pragma unroll
                for (int xx = 0; xx < kTex; xx++) {
                    const int lx = (int)x, ly = (int)y;
                    wx[xx] = x - (float)lx; wy[xx] = y - (float)ly;
                    toff[xx] = 3 * (ly * W + lx);
                    ta[xx] = make_uint2(0u, 0u); tb[xx] = make_uint2(0u, 0u);
                    x += dxx; y += dxy;
                }
                if (act) {
#pragma unroll
                    for (int xx = 0; xx < kTex; xx++) {
                        const uint8_t* p0 = pix + toff[xx];
                        ta[xx] = load8(p0);
                        tb[xx] = load8(p0 + 3 * W);
                    }
                }
                // Image::getColor two texels at a time: every float32 operation of texel() on the two halves of a
                // register pair (v_pk_mul_f32 / v_pk_add_f32 are IEEE per half: same values, half the instructions)
#pragma unroll
                for (int xx = 0; xx + 1 < kTex; xx += 2)
                    texel_pair(ta[xx], tb[xx], wx[xx], wy[xx], ta[xx + 1], tb[xx + 1], wx[xx + 1], wy[xx + 1], &t[3 * xx], &t[3 * xx + 3]);
                {
                    const int xx = kTex - 1;
                    Taps tp;
                    tp.a = ta[xx]; tp.b = tb[xx];
                    const float dx1 = wx[xx], dx0 = 1.0f - dx1;
                    const float dy1 = wy[xx], dy0 = 1.0f - dy1;
                    tp.f00 = dx0 * dy0; tp.f01 = dx0 * dy1; tp.f10 = dx1 * dy0; tp.f11 = dx1 * dy1;
                    texel(tp, t[3 * xx], t[3 * xx + 1], t[3 * xx + 2]);
                }
            }
            PH_T(t4);
            PH_ADD(t3, t4, 14);

            // ---- 4. Patch2d::normalize (Patch2d.hpp:46-84): channel means, variance, normalisation
            float a0, a1, a2;
            {
                float vr[kTex], vg[kTex], vb[kTex];
#pragma unroll
                for (int xx = 0; xx < kTex; xx++) { vr[xx] = t[3 * xx]; vg[xx] = t[3 * xx + 1]; vb[xx] = t[3 * xx + 2]; }
                float s0, s1, s2;
                mean_chains(vr, vg, vb, s0, s1, s2);
                a0 = div_by(shfl_f(s0, last), 1.0 / 49.0);
                a1 = div_by(shfl_f(s1, last), 1.0 / 49.0);
                a2 = div_by(shfl_f(s2, last), 1.0 / 49.0);
            }
            double iv;
            {
                float sq[kTex];
#pragma unroll
                for (int xx = 0; xx < kTex; xx++) {
                    const float f0 = a0 - t[3 * xx], f1 = a1 - t[3 * xx + 1], f2 = a2 - t[3 * xx + 2];
                    sq[xx] = f0 * f0 + f1 * f1 + f2 * f2;
                }
                float v2 = sqrtf(div_by(shfl_f(window_sum<kTex>(sq), last), 1.0 / 147.0));
                if (v2 == 0.0f) v2 = 1.0f;
                iv = recip64(v2);
            }
            PH_T(t5);
            PH_ADD(t4, t5, 15);
#pragma unroll
            for (int xx = 0; xx < kTex; xx++) {
                t[3 * xx] = div_by(t[3 * xx] - a0, iv);
                t[3 * xx + 1] = div_by(t[3 * xx + 1] - a1, iv);
                t[3 * xx + 2] = div_by(t[3 * xx + 2] - a2, iv);
            }
            // ---- 5. Patch2d::dot with the request's reference texture (Patch2d.hpp:37-44)
            float raw = 0.0f, rob = 0.0f;
#pragma unroll
            for (int i = 0; i < kRowVals; i++) refn[i] = g_isref ? t[i] : refn[i];
            if (moved) {  // (wave-uniform) the helpers of this pass fetch their victims' normalised reference texels, row by row --
                          // also those a victim has only just sampled: a request's reference image and its other images run side by side
#pragma unroll
                for (int i = 0; i < kRowVals; i++) refn[i] = shfl_f(refn[i], rsrc);
            }
            PH_T(t6);
            PH_ADD(t5, t6, 20);
            if (__any(act && !g_isref)) {  // (wave-uniform) the first pass of a batch is all reference grabs: nothing to correlate
                float pr[kRowVals];
#pragma unroll
                for (int i = 0; i < kRowVals; i++) pr[i] = refn[i] * t[i];
                raw = 1.0f - div_by(shfl_f(window_sum<kRowVals>(pr), last), 1.0 / 147.0);
                rob = robustincc(raw);
            }
            PH_T(t7);
            PH_ADD(t6, t7, 21);

            // ---- 6. the grab's result goes to its place in the request's row; next image of the request, or idle
            if (act) {
                if (!g_isref && yy == 0) rows.store(g_req, g_cur, g_userob ? rob : raw);
                if (g_ok) { g_cur = __ffsll((long long)g_ok) - 1; g_ok &= g_ok - 1ull; g_isref = false; }
                else g_req = -1;
            }
            PH_T(t8);
            PH_ADD(t7, t8, 22);
        }

        // ---- 7. the reference's accumulations, in index order, on the requesting lanes (lane-parallel over the batch's
        //         requests): objective_fn (PatchOptimizer.cpp:299-309), setINCCs + filterImagesNCC / std::accumulate
        //         (:462-472, :144-151, :161-167).  A failed grab holds 2.0, the reference image 0 (adding it changes
        //         nothing: the running sums are never -0).
        PH_T(t9);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the groups' stores are out
        __builtin_amdgcn_wave_barrier();
        if (inbatch && !refok) {  // sampleTexture false for the reference image: plateau value / all 2.0
            if (q.obj_mode) out.obj = 2.0;
            else {
                float s = 0.0f;
                for (int k = 0; k < q.nimg; k++) s += 2.0f;
                out.keep = 0ull;
                out.incc_sum = s;
            }
        }
        {
            const bool fin = inbatch && refok;
            const float thr1 = 1.0f - q.thr;
            double val = 0.0;
            int nok = 0;
            float s = 0.0f;
            unsigned long long keep = 0ull;
            const int kmax = rows.stride - 1;
            for (int k0 = 0; __any(fin && k0 < q.nimg); k0 += kFinChunk) {
                float e[kFinChunk];
#pragma unroll
                for (int j = 0; j < kFinChunk; j++) e[j] = 2.0f;
                if (fin && k0 < q.nimg) {
#pragma unroll
                    for (int j = 0; j < kFinChunk; j++) { const int kk = k0 + j < kmax ? k0 + j : kmax; e[j] = rows.load(lane, kk); }
                }
#pragma unroll
                for (int j = 0; j < kFinChunk; j++) {
                    const int k = k0 + j;
                    const bool in = fin && k < q.nimg;
                    const bool okk = (myok >> (k & 63)) & 1ull;
                    const bool isref = k == q_refsel;
                    const unsigned long long bit = 1ull << (k & 63);
                    // objective_fn: the usable images other than the reference
                    const bool add_o = in && okk && !isref;
                    val += add_o ? (double)e[j] : 0.0;
                    nok += add_o ? 1 : 0;
                    // setINCCs: every image; inccs[ref] = 0
                    const float v = okk ? e[j] : 2.0f;
                    s += (in && !isref) ? v : 0.0f;
                    const bool kp = in && (isref ? (0.0f < thr1) : (v < thr1));
                    keep |= kp ? bit : 0ull;
                }
            }
            if (fin) {
                out.ngrabs += mylen;
                if (q.obj_mode) out.obj = (nok < o.MIN_IMAGES_PER_PATCH - 1) ? 2.0 : val / (double)nok;
                else {
                    out.keep = WRITE_INCCS ? myok : keep;
                    out.incc_sum = s;
                    if (WRITE_INCCS) rows.store(lane, q_refsel, 0.0f);
                }
            }
        }
        PH_T(t10);
        PH_ADD(t9, t10, 16);
    }
}


// ------------------------------------------------------------------------------------------
// optimiser-variable <-> patch maps (per-lane scalar)

struct OptFields {       // setOptimizationFields (PatchOptimizer.cpp:384-399)
    float refCenter[3];  // w = 1
    float refRay[3];     // w = 0
    float imgX[3], imgY[3], imgZ[3];  // axes of the reference image's camera, normalised
};

__device__ __forceinline__ void set_optimization_fields(const DevView& V0, const float* c, OptFields& F) {
    normalized3f(V0.xaxis, F.imgX);
    normalized3f(V0.yaxis, F.imgY);
    normalized3f(V0.zaxis, F.imgZ);
    F.refCenter[0] = c[0]; F.refCenter[1] = c[1]; F.refCenter[2] = c[2];
    const float d[3] = {c[0] - V0.center[0], c[1] - V0.center[1], c[2] - V0.center[2]};
    normalized3f(d, F.refRay);
}

// sin and cos of a float64 angle in [-pi/2 - eps, pi/2 + eps] (the optimiser bounds the angles to
// +-23.99999 * pi/48), Taylor series in Horner form: truncation < 2e-18, a couple of ulp of rounding.
// The reference calls the C library's double sin/cos on the float angle (unqualified sin(float) with
// only <cmath> in scope) and rounds the products to float, which absorbs the last-ulp differences.
__device__ __forceinline__ void sincos_halfpi(double a, double& s, double& c) {
    const double z = a * a;
    double ps = -1.0 / 51090942171709440000.0;                 // -1/21!
    ps = ps * z + 1.0 / 121645100408832000.0;                  //  1/19!
    ps = ps * z - 1.0 / 355687428096000.0;                     // -1/17!
    ps = ps * z + 1.0 / 1307674368000.0;                       //  1/15!
    ps = ps * z - 1.0 / 6227020800.0;                          // -1/13!
    ps = ps * z + 1.0 / 39916800.0;                            //  1/11!
    ps = ps * z - 1.0 / 362880.0;                              // -1/9!
    ps = ps * z + 1.0 / 5040.0;                                //  1/7!
    ps = ps * z - 1.0 / 120.0;                                 // -1/5!
    ps = ps * z + 1.0 / 6.0;                                   //  1/3!  (sign folded below)
    s = a - a * z * ps;
    double pc = 1.0 / 1124000727777607680000.0;                //  1/22!
    pc = pc * z - 1.0 / 2432902008176640000.0;                 // -1/20!
    pc = pc * z + 1.0 / 6402373705728000.0;                    //  1/18!
    pc = pc * z - 1.0 / 20922789888000.0;                      // -1/16!
    pc = pc * z + 1.0 / 87178291200.0;                         //  1/14!
    pc = pc * z - 1.0 / 479001600.0;                           // -1/12!
    pc = pc * z + 1.0 / 3628800.0;                             //  1/10!
    pc = pc * z - 1.0 / 40320.0;                               // -1/8!
    pc = pc * z + 1.0 / 720.0;                                 //  1/6!
    pc = pc * z - 1.0 / 24.0;                                  // -1/4!
    pc = pc * z + 0.5;                                         //  1/2!  (sign folded below)
    c = 1.0 - z * pc;
}

// setCenterNorm (PatchOptimizer.cpp:401-414); depthScale_ = 1
__device__ __forceinline__ void set_center_norm(const OptFields& F, const DevOptions& o, const double* x, float* c,
                                                float* n) {
    const float x0 = (float)x[0];
    for (int k = 0; k < 3; k++) c[k] = F.refCenter[k] + (x0 * F.refRay[k]) * 1.0f;
    const float angle1 = (float)(x[1] * (double)o.angle_scale);
    const float angle2 = (float)(x[2] * (double)o.angle_scale);
    double s1, c1, s2, c2;
    sincos_halfpi((double)angle1, s1, c1);
    sincos_halfpi((double)angle2, s2, c2);
    const float fx = (float)(s1 * c2);
    const float fy = (float)s2;
    const float fz = (float)(-c1 * c2);
    for (int k = 0; k < 3; k++) n[k] = (F.imgX[k] * fx + F.imgY[k] * fy) + F.imgZ[k] * fz;
}

// std::asin(float) as the host C library computes it.  x[2] of the BOBYQA start point is
// asinf(fy) / angleScale (PatchOptimizer.cpp:427): a one-ulp float difference there changes the
// whole trajectory, so the device evaluates the SAME algorithm as glibc 2.35's __ieee754_asinf
// (sysdeps/ieee754/flt-32/e_asinf.c: 5-term polynomial on |x| < 0.5, sqrt reduction with a split
// high part above), float32 operations in the same order.  tools/asinf_check.py verifies the
// restatement bit for bit against the host libm.
__device__ __forceinline__ float asinf_glibc(float x) {
    const float one = 1.0f, pio2_hi = 1.57079637050628662109375f, pio2_lo = -4.37113900018624283e-8f,
                pio4_hi = 0.785398185253143310546875f;
    const float p0 = 1.666675248e-1f, p1 = 7.495297643e-2f, p2 = 4.547037598e-2f, p3 = 2.417951451e-2f,
                p4 = 4.216630880e-2f;
    const int hx = __float_as_int(x);
    const int ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return x * pio2_hi + x * pio2_lo;
    if (ix > 0x3f800000) return (x - x) / (x - x);
    if (ix < 0x3f000000) {
        if (ix < 0x32000000) return x;
        const float t = x * x;
        const float w = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
        return x + x * w;
    }
    float w = one - fabsf(x);
    float t = w * 0.5f;
    float p = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
    const float s = sqrtf(t);
    if (ix >= 0x3F79999A) {
        t = pio2_hi - (2.0f * (s + s * p) - pio2_lo);
    } else {
        w = __int_as_float(__float_as_int(s) & 0xfffff000);
        const float c = (t - w * w) / (s + w);
        const float r = p;
        p = 2.0f * s * r - (pio2_lo - 2.0f * c);
        const float q = pio4_hi - 2.0f * w;
        t = pio4_hi - (p - q);
    }
    return hx > 0 ? t : -t;
}

// parametersFromCenterNorm (PatchOptimizer.cpp:416-446) for c = refCenter_ (the only call site)
__device__ __forceinline__ void parameters_from_center_norm(const OptFields& F, const DevOptions& o, const float* n,
                                                            const double* lb, const double* ub, double* x) {
    x[0] = 0.0;  // (refCenter - refCenter) . refRay
    const float fx = dot3f(F.imgX, n), fy = dot3f(F.imgY, n), fz = dot3f(F.imgZ, n);
    x[2] = (double)asinf_glibc(fy);
    double t = (x[2] < 1.0) ? x[2] : 1.0;
    t = (-1.0 < t) ? t : -1.0;
    const float cosb = (float)cos(t);
    if (cosb == 0.0f) x[1] = 0.0;
    else {
        const double sina = (double)(fx / cosb);
        const double cosa = (double)(-fz / cosb);
        double u = (-1.0 < cosa) ? cosa : -1.0;
        u = (u < 1.0) ? u : 1.0;
        x[1] = cra::acos_cr(u, acos(u));  // rounded to nearest like the host libm (cr_acos.hpp)
        if (sina < 0.0) x[1] = -x[1];
    }
    x[1] /= (double)o.angle_scale;
    x[2] /= (double)o.angle_scale;
    for (int k = 0; k < 3; k++) {
        const double v = (lb[k] < x[k]) ? x[k] : lb[k];
        x[k] = (v < ub[k]) ? v : ub[k];
    }
}


}  // namespace hpmvs
