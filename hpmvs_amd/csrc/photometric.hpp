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
//   prep_image, issue_taps,
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
    if (r2 <= 0.0f) { u = -65535.0f; v = -65535.0f; return; }
    const float lo = -2147483648.0f, hi = 2147483648.0f;  // (float)(INT_MIN+3.0f), (float)(INT_MAX-3.0f)
    u = fmaxf(lo, fminf(hi, r0 / r2));
    v = fmaxf(lo, fminf(hi, r1 / r2));
}

// Camera::getLeveli (Camera.cpp:92-99): clamp(round(log2(scale*fsum/(2*fz))), 0, max_level),
// evaluated as a count of thresholds r >= 2^(k+1/2) in float64 (no log, no division).
__device__ __forceinline__ int level_clamped(float scale, float fsum, float fz, int max_level) {
    const double num = (double)(scale * fsum);
    const double den = 2.0 * (double)fz;
    int lvl = 0;
    double t = 1.4142135623730951;  // 2^(0+1/2)
    for (int k = 0; k < kMaxLevels - 1; k++) {
        if (k < max_level && num >= t * den) lvl++;
        t *= 2.0;
    }
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

// The per-image, patch-uniform part of sampleTexture (PatchOptimizer.cpp:476-508): angle gate,
// pyramid level, projected window, bounding-box gate.
__device__ __forceinline__ Prep prep_image(const DevView& V, const DevOptions& o, const float* c, float scale,
                                           const float* pxa, const float* pya, const float* zax) {
    Prep r;
    r.ok = 0; r.tlx = r.tly = r.dxx = r.dxy = r.dyx = r.dyy = 0.0f; r.pix = nullptr; r.W = 0;
    const float d0 = V.center[0] - c[0], d1 = V.center[1] - c[1], d2 = V.center[2] - c[2];
    const float n2 = (d0 * d0 + d1 * d1) + d2 * d2;  // w components cancel (1 - 1)
    const float nn = sqrtf(n2);
    float g;
    if (n2 > 0.0f) g = ((d0 / nn) * zax[0] + (d1 / nn) * zax[1]) + (d2 / nn) * zax[2];
    else g = (d0 * zax[0] + d1 * zax[1]) + d2 * zax[2];
    if ((double)g < o.cos_max_d) return r;

    const int lvl = level_clamped(scale, V.fsum, nn, o.MAXLEVEL - 1);
    float P[12];
#pragma unroll
    for (int k = 0; k < 12; k++) P[k] = V.P0[k];
    float cu, cv, xu, xv, yu, yv;
    project0(P, c[0], c[1], c[2], cu, cv);
    project0(P, c[0] + pxa[0], c[1] + pxa[1], c[2] + pxa[2], xu, xv);
    project0(P, c[0] + pya[0], c[1] + pya[1], c[2] + pya[2], yu, yv);
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
    const int W = V.w[lvl], H = V.h[lvl];
    const float m = 3.0f;
    if (mnx < m || mny < m || mxx >= (float)(W - 3) || mxy >= (float)(H - 3)) return r;
    r.ok = 1;
    r.tlx = tlx; r.tly = tly; r.dxx = dxx; r.dxy = dxy; r.dyx = dyx; r.dyy = dyy;
    r.pix = V.pix[lvl];
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

struct EvalResult {
    double obj;       // OBJ: objective_fn value
    float incc_lane;  // INCC: lane v holds inccs[v]
    float incc_sum;   // INCC: sequential float sum of inccs (std::accumulate, PatchOptimizer.cpp:165)
    int ngrabs;       // grabs that passed the gates
};

// ------------------------------------------------------------------------------------------
// EXACT-ORDER evaluator.
//
// BOBYQA on this objective amplifies a 1e-9 relative perturbation of f into a different basin for
// 7-8 % of patches (tests/sensitivity_probe.py), so "close" objective values are not good enough
// for parity: the objective has to be the reference's float32 arithmetic bit for bit, including
// the SEQUENTIAL summation order of Patch2d::normalize / dot (Patch2d.hpp:37-84) and the
// incremental sample stepping of sampleTexture (PatchOptimizer.cpp:511-526).  A wave reduction
// cannot do that, but the sums of DIFFERENT images are independent, so:
//   1. lanes 0..48 fetch the raw texels of up to kChunk images (all tap loads of a chunk are issued
//      before any is consumed) and stage them in LDS in the reference's data[(y*7+x)*3+c] layout;
//   2. lane (m*3+c) forms the channel mean of image m sequentially over the 49 pixels;
//   3. lane m forms the variance sum sequentially; 4. lanes 0..48 normalise in place;
//   5. lane m forms the 147-term dot product with the reference texture sequentially.
// Built with -ffp-contract=off, every float operation is the one the CPU executes.
#ifndef HPMVS_CHUNK
#define HPMVS_CHUNK 7
#endif
constexpr int kChunk = HPMVS_CHUNK;  // images per chunk besides the reference texture
constexpr int kPlane = 52;           // floats per staged channel plane (49 + 3: 16-byte aligned planes)
constexpr int kTexStride = 164;      // floats per staging slot (16-byte multiple; 164 mod 64 = 36 spreads the per-slot float4 reads over distinct LDS banks): 3 planes of raw texels, later reused for the
                                     // 49 squared deviations and then the 147 products ref[k]*img[k]
struct alignas(16) EvalLds {
    float buf[(kChunk + 1) * kTexStride];  // slot 0: reference image, slots 1..kChunk: chunk members
    float ave[(kChunk + 1) * 4];           // per slot: channel means, then sqrt(variance)
    double inv[kChunk + 1];                // per slot: 1 / sqrt(variance) in float64 (see div_by)
};

__device__ __forceinline__ void wave_lds_sync() {
    // single wavefront: DS operations execute in issue order; this only stops the compiler from
    // moving LDS accesses across the hand-over between lanes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sequential float sum of N contiguous LDS floats (16-byte aligned, padded to a multiple of 4) by one
// lane: float4 reads, adds strictly in index order
template <int N>
__device__ __forceinline__ float seq_sum(const float* p) {
    const float4* p4 = reinterpret_cast<const float4*>(p);
    float a = 0.0f;
    constexpr int Q = N / 4;
#pragma unroll 6
    for (int q = 0; q < Q; q++) {
        const float4 u = p4[q];
        a += u.x; a += u.y; a += u.z; a += u.w;
    }
    if (N % 4) {
        const float4 u = p4[Q];
        if (N % 4 >= 1) a += u.x;
        if (N % 4 >= 2) a += u.y;
        if (N % 4 >= 3) a += u.z;
    }
    return a;
}

struct Taps { uint2 a, b; float f00, f01, f10, f11; };

// address + weights of the texel this lane samples (sampleTexture's stepping: l += dy per row,
// c += dx per column, PatchOptimizer.cpp:511-526), and the two tap loads
__device__ __forceinline__ Taps issue_taps(float tlx, float tly, float dxx, float dxy, float dyx, float dyy,
                                           const uint8_t* pix, int W, int ixx, int iyy) {
    // x += step taken `iyy` / `ixx` times: fma(m, step, x) with m in {0, 1} is the same single rounding
    // as x + step (m*step is exact) and leaves x untouched for m = 0
    float x = tlx, y = tly;
#pragma unroll
    for (int k = 0; k < kTex - 1; k++) { const float m = (k < iyy) ? 1.0f : 0.0f; x = __builtin_fmaf(m, dyx, x); y = __builtin_fmaf(m, dyy, y); }
#pragma unroll
    for (int k = 0; k < kTex - 1; k++) { const float m = (k < ixx) ? 1.0f : 0.0f; x = __builtin_fmaf(m, dxx, x); y = __builtin_fmaf(m, dxy, y); }
    const int lx = (int)x, ly = (int)y;
    const uint8_t* p0 = pix + 3 * (ly * W + lx);
    Taps t;
    t.a = load8(p0);
    t.b = load8(p0 + 3 * W);
    const float dx1 = x - (float)lx, dx0 = 1.0f - dx1;
    const float dy1 = y - (float)ly, dy0 = 1.0f - dy1;
    t.f00 = dx0 * dy0; t.f01 = dx0 * dy1; t.f10 = dx1 * dy0; t.f11 = dx1 * dy1;
    return t;
}

// Image::getColor (Image.h:104-113) in its evaluation order
__device__ __forceinline__ void texel(const Taps& t, float& r, float& g, float& b) {
    r = 0.0f; g = 0.0f; b = 0.0f;
    r += (float)(t.a.x & 0xff) * t.f00 + (float)(t.b.x & 0xff) * t.f01;
    g += (float)((t.a.x >> 8) & 0xff) * t.f00 + (float)((t.b.x >> 8) & 0xff) * t.f01;
    b += (float)((t.a.x >> 16) & 0xff) * t.f00 + (float)((t.b.x >> 16) & 0xff) * t.f01;
    r += (float)(t.a.x >> 24) * t.f10 + (float)(t.b.x >> 24) * t.f11;
    g += (float)(t.a.y & 0xff) * t.f10 + (float)(t.b.y & 0xff) * t.f11;
    b += (float)((t.a.y >> 8) & 0xff) * t.f10 + (float)((t.b.y >> 8) & 0xff) * t.f11;
}

// px, py, pz: calculatePatchAxis output for the reference image's camera (wave-uniform, computed by
// the requesting lane in phase A / by the caller).
template <typename ListT>
__device__ __forceinline__ EvalResult wave_eval_exact(const DevScene& sc, const DevOptions& o, const ListT* list,
                                                      int nimg, int ref_slot, const float* c, const float* n,
                                                      float scale, const float* px, const float* py, const float* pz,
                                                      int obj_mode, int robust, EvalLds& L) {
    EvalResult res;
    res.obj = 2.0; res.incc_lane = 2.0f; res.incc_sum = 0.0f; res.ngrabs = 0;
    const int lane = lane_id();
    const bool active = lane < kTexPix;
    const int sl = active ? lane : kTexPix - 1;
    const int iyy = sl / kTex, ixx = sl % kTex;
    PH_T(t1);
    float zax[3];
    for (int k = 0; k < 3; k++) zax[k] = obj_mode ? pz[k] : n[k];

    Prep P;
    P.ok = 0; P.tlx = P.tly = P.dxx = P.dxy = P.dyx = P.dyy = 0.0f; P.pix = nullptr; P.W = 0;
    if (lane < nimg) P = prep_image(sc.views[(int)list[lane]], o, c, scale, px, py, zax);
    PH_T(t2);
    PH_ADD(t1, t2, 13);

    const unsigned long long okmask = __ballot(P.ok != 0);
    if (!((okmask >> ref_slot) & 1ull)) {
        if (!obj_mode) {
            float s = 0.0f;
            for (int k = 0; k < nimg; k++) s += 2.0f;
            res.incc_sum = s;
        }
        return res;
    }
    unsigned long long todo = okmask & ~(1ull << ref_slot);
    int ngr = 1, nok = 0;
    double val = 0.0;
    float mine = (lane == ref_slot) ? 0.0f : 2.0f;
    float refn0 = 0.0f, refn1 = 0.0f, refn2 = 0.0f;  // this lane's normalised reference texel
    bool first = true;
    while (first || todo) {
        // ---- members of this chunk: the lowest set bits of `todo`, ascending image index
        unsigned long long chunk = 0ull, t = todo;
        for (int k = 0; k < kChunk && t; k++) { chunk |= t & (~t + 1ull); t &= t - 1ull; }
        todo = t;
        const int nm = __popcll(chunk);
        // lane s in 1..nm knows which image it stands for; slot 0 is the reference image
        int member = ref_slot;
        if (lane >= 1 && lane <= nm) {
            unsigned long long u = chunk;
            for (int j = 1; j < lane; j++) u &= u - 1ull;
            member = __ffsll((long long)u) - 1;
        }
        const int s0 = first ? 0 : 1;
        PH_T(t3);
        // ---- 1. fetch: issue every tap load of the chunk, then convert; stage the raw texels as
        //         three channel planes per slot (plane c at buf[c*kPlane + pixel])
        Taps taps[kChunk + 1];
        float rr[kChunk + 1], gg[kChunk + 1], bb[kChunk + 1];
#pragma unroll
        for (int s = 0; s <= kChunk; s++) {
            if (s >= s0 && s <= nm) {
                const int v = bcast_i(member, s);
                taps[s] = issue_taps(bcast_f(P.tlx, v), bcast_f(P.tly, v), bcast_f(P.dxx, v), bcast_f(P.dxy, v),
                                     bcast_f(P.dyx, v), bcast_f(P.dyy, v), bcast_p(P.pix, v), bcast_i(P.W, v), ixx, iyy);
            }
        }
#pragma unroll
        for (int s = 0; s <= kChunk; s++) {
            rr[s] = gg[s] = bb[s] = 0.0f;
            if (s >= s0 && s <= nm) {
                texel(taps[s], rr[s], gg[s], bb[s]);
                if (active) {
                    float* bp = &L.buf[s * kTexStride];
                    bp[lane] = rr[s]; bp[kPlane + lane] = gg[s]; bp[2 * kPlane + lane] = bb[s];
                }
            }
        }
        wave_lds_sync();
        PH_T(t4);
        PH_ADD(t3, t4, 14);

        // ---- 2. channel means (Patch2d.hpp:51-58): lane m*3+c adds the 49 texels of plane c of slot m in order
        {
            const int m = lane / 3, ch = lane - 3 * m;
            if (m >= s0 && m <= nm) L.ave[m * 4 + ch] = div_by(seq_sum<49>(&L.buf[m * kTexStride + ch * kPlane]), 1.0 / 49.0);
        }
        wave_lds_sync();
        // ---- 3. squared deviations per pixel (Patch2d.hpp:62-69), computed by the pixel lanes ...
#pragma unroll
        for (int s = 0; s <= kChunk; s++) {
            if (s >= s0 && s <= nm && active) {
                const float4 av = *reinterpret_cast<const float4*>(&L.ave[s * 4]);
                const float f0 = av.x - rr[s], f1 = av.y - gg[s], f2 = av.z - bb[s];
                L.buf[s * kTexStride + lane] = f0 * f0 + f1 * f1 + f2 * f2;
            }
        }
        wave_lds_sync();
        // ---- ... and summed in pixel order by lane m
        if (lane >= s0 && lane <= nm) {
            float v2 = sqrtf(div_by(seq_sum<49>(&L.buf[lane * kTexStride]), 1.0 / 147.0));
            if (v2 == 0.0f) v2 = 1.0f;
            L.inv[lane] = recip64(v2);
        }
        wave_lds_sync();
        PH_T(t5);
        PH_ADD(t4, t5, 15);

        // ---- 4. normalise (Patch2d.hpp:75-83) in registers; products with the reference texture
        //         (Patch2d.hpp:41) staged in the reference's element order (pixel*3 + channel)
#pragma unroll
        for (int s = 0; s <= kChunk; s++) {
            if (s >= s0 && s <= nm) {
                const float4 av = *reinterpret_cast<const float4*>(&L.ave[s * 4]);
                float r = rr[s], g = gg[s], b = bb[s];
                const double iv = L.inv[s];
                r -= av.x; r = div_by(r, iv);
                g -= av.y; g = div_by(g, iv);
                b -= av.z; b = div_by(b, iv);
                if (s == 0) { refn0 = r; refn1 = g; refn2 = b; }
                else if (active) {
                    float* bp = &L.buf[s * kTexStride + 3 * lane];
                    bp[0] = refn0 * r; bp[1] = refn1 * g; bp[2] = refn2 * b;
                }
            }
        }
        wave_lds_sync();
        PH_T(t6);
        PH_ADD(t5, t6, 20);

        // ---- 5. NCC: lane m adds the 147 products in order (Patch2d.hpp:37-44)
        float raw = 2.0f;
        if (lane >= 1 && lane <= nm) raw = 1.0f - div_by(seq_sum<147>(&L.buf[lane * kTexStride]), 1.0 / 147.0);
        PH_T(t7);
        PH_ADD(t6, t7, 21);

        // ---- deliver in ascending image order
        for (int s = 1; s <= nm; s++) {
            const int v = bcast_i(member, s);
            const float e = bcast_f(raw, s);
            ngr++;
            if (obj_mode) {
                val += (double)robustincc(e);
                nok++;
            } else if (lane == v) {
                mine = robust ? robustincc(e) : e;
            }
        }
        wave_lds_sync();
        first = false;
        PH_T(t8);
        PH_ADD(t7, t8, 22);
    }
    res.ngrabs = ngr;
    if (obj_mode) {
        res.obj = (nok < o.MIN_IMAGES_PER_PATCH - 1) ? 2.0 : val / (double)nok;
    } else {
        res.incc_lane = mine;
        float s = 0.0f;
        for (int k = 0; k < nimg; k++) s += bcast_f(mine, k);
        res.incc_sum = s;
    }
    return res;
}

// ------------------------------------------------------------------------------------------
// PACKED exact-order evaluator: every request a wave has pending, in one pass.
//
// wave_eval_exact serves ONE request and leaves most lanes idle: a typical objective_fn request has
// 3-4 usable images, i.e. 3-4 of the 8 staging slots, 3-4 lanes on the sequential sums, 3-4 lanes of
// prep_image.  All of that is latency, not throughput (profiles/r02: 17 k cycles per request with
// the wave alone on its CU).  Here the (request, image) pairs of ALL pending requests become "jobs":
//   1. one lane per job runs prep_image (request data fetched from the requesting lane with
//      ds_bpermute);
//   2. the usable jobs are laid out as one sequence -- per request: reference image first, then the
//      other images in ascending index -- and cut into chunks of 8 staging slots, regardless of
//      request boundaries (the normalised reference texel and the per-request accumulators live in
//      registers and simply persist across a cut);
//   3. each chunk runs the same five steps as wave_eval_exact (fetch+stage, channel means, variance,
//      normalise+products, 147-term sum); results are delivered per request in image order.
// Per image the float32 operations and their order are exactly those of wave_eval_exact (and of the
// reference); only the interleaving of independent images changes.
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

__device__ __forceinline__ float shfl_f(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}
__device__ __forceinline__ int shfl_i(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }

// end of a setINCCs request: sequential float sum of the per-image values (lane k holds inccs[k]) and the
// filterImagesNCC mask; out of line so that the unrolled delivery loop carries one copy
__device__ __noinline__ void incc_finalize(float mine, int np, float thr, unsigned long long& keep, float& sum) {
    float s = 0.0f;
    for (int k = 0; k < np; k++) s += bcast_f(mine, k);
    sum = s;
    keep = __ballot(mine < 1.0f - thr);
}

// image id `i` of the list at `row`.  COHERENT: the list was written by ANOTHER wavefront during this launch
// (stream kernel: write-through sc1 stores), so the read bypasses this CU's L1 (sc1 load, 4-byte granule).
template <bool COHERENT>
__device__ __forceinline__ int list_id(const uint16_t* row, int i) {
    if (COHERENT) {
        const unsigned int w = __hip_atomic_load((const unsigned int*)(row + (i & ~1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (i & 1) ? (int)(w >> 16) : (int)(w & 0xffffu);
    }
    return (int)row[i];
}

// myrow: the calling lane's own image list (uint16 ids, 4-byte aligned; read only for lanes with a pending
// request).  Must be called by all 64 lanes.
template <bool COHERENT>
__device__ __forceinline__ void wave_eval_packed(const DevScene& sc, const DevOptions& o, const uint16_t* myrow,
                                                 unsigned long long pending, const PackedReq& q, PackedRes& out,
                                                 EvalLds& L) {
    const unsigned long long rowbits_ = (unsigned long long)myrow;
    const int row_lo_ = (int)(rowbits_ & 0xffffffffull), row_hi_ = (int)(rowbits_ >> 32);
    constexpr int S = kChunk + 1;  // staging slots
    const int lane = lane_id();
    const bool active = lane < kTexPix;
    const int sl = active ? lane : kTexPix - 1;
    const int iyy = sl / kTex, ixx = sl % kTex;
    const int q_refsel = q.obj_mode ? 0 : q.ref;

    unsigned long long rem = pending;
    while (rem) {
        // ---- a batch of requests whose jobs fit the 64 lanes (at least one request)
        unsigned long long batch = 0ull;
        int J = 0;
        for (unsigned long long t = rem; t; t &= t - 1ull) {
            const int p = __ffsll((long long)t) - 1;
            const int np = bcast_i(q.nimg, p);
            if (batch && J + np > 64) break;
            batch |= 1ull << p;
            J += np;
        }
        rem &= ~batch;
        PH_T(t1);
        // ---- job of this lane: (request myp, image index myi)
        int myp = -1, myi = 0;
        {
            int off = 0;
            for (unsigned long long t = batch; t; t &= t - 1ull) {
                const int p = __ffsll((long long)t) - 1;
                const int np = bcast_i(q.nimg, p);
                if (lane >= off && lane < off + np) { myp = p; myi = lane - off; }
                off += np;
            }
        }
        const int src = myp >= 0 ? myp : 0;
        float c[3], zax[3], pxa[3], pya[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            c[k] = shfl_f(q.c[k], src); zax[k] = shfl_f(q.zax[k], src);
            pxa[k] = shfl_f(q.px[k], src); pya[k] = shfl_f(q.py[k], src);
        }
        const float scale = shfl_f(q.scale, src);
        Prep P;
        P.ok = 0; P.tlx = P.tly = P.dxx = P.dxy = P.dyx = P.dyy = 0.0f; P.pix = nullptr; P.W = 0;
        const unsigned long long rb_ = ((unsigned long long)(unsigned int)shfl_i(row_hi_, src) << 32) |
                                       (unsigned long long)(unsigned int)shfl_i(row_lo_, src);
        if (myp >= 0) P = prep_image(sc.views[list_id<COHERENT>((const uint16_t*)rb_, myi)], o, c, scale, pxa, pya, zax);
        const unsigned long long okmask = __ballot(P.ok != 0);
        PH_T(t2);
        PH_ADD(t1, t2, 13);

        // ---- position of every usable job in the slot sequence
        int pos = -1;
        bool isref = false;
        int E = 0;
        unsigned long long refok = 0ull;  // requests whose reference image passed the gates
        {
            int off = 0;
            for (unsigned long long t = batch; t; t &= t - 1ull) {
                const int p = __ffsll((long long)t) - 1;
                const int np = bcast_i(q.nimg, p);
                const int refp = bcast_i(q_refsel, p);
                if (np > 0 && ((okmask >> (off + refp)) & 1ull)) {
                    refok |= 1ull << p;
                    const unsigned long long R = (np >= 64 ? ~0ull : ((1ull << np) - 1ull)) << off;
                    const unsigned long long okr = okmask & R;
                    const unsigned long long others = okr & ~(1ull << (off + refp));
                    if (myp == p && P.ok) {
                        if (myi == refp) { pos = E; isref = true; }
                        else pos = E + 1 + __popcll(others & ((1ull << lane) - 1ull));
                    }
                    E += __popcll(okr);
                }
                off += np;
            }
        }
        const unsigned long long refjobs = __ballot(isref);

        // ---- per-request running state (persists across chunk cuts)
        int cur = -1, cur_obj = 0, cur_rob = 0, nok = 0, ngr = 0;
        double val = 0.0;
        float mine = 2.0f;
        float refn0 = 0.0f, refn1 = 0.0f, refn2 = 0.0f;
        // OBJ results are parked on the requesting lane and divided once, lane-parallel, after the last chunk
        double d_val = 0.0;
        int d_nok = -1;
        auto finalize = [&]() {
            if (cur_obj) {
                if (lane == cur) { d_val = val; d_nok = nok; out.ngrabs += ngr; }
            } else {
                unsigned long long keep;
                float s;
                incc_finalize(mine, bcast_i(q.nimg, cur), bcast_f(q.thr, cur), keep, s);
                if (lane == cur) { out.keep = keep; out.incc_sum = s; out.ngrabs += ngr; }
            }
        };

        for (int cb = 0; cb < E; cb += S) {
            const int ns = (E - cb < S) ? E - cb : S;
            PH_T(t3);
            // job lane of every slot (wave-uniform), also kept per lane for the dynamic delivery loop
            int jl[S];
            unsigned int refslots = 0u;
#pragma unroll
            for (int s = 0; s < S; s++) {
                const unsigned long long m = __ballot(pos == cb + s);
                jl[s] = m ? __ffsll((long long)m) - 1 : 0;
                if (s < ns && ((refjobs >> jl[s]) & 1ull)) refslots |= 1u << s;
            }
            // ---- 1. fetch + stage
            Taps taps[S];
            float rr[S], gg[S], bb[S];
#pragma unroll
            for (int s = 0; s < S; s++) {
                if (s < ns) {
                    const int v = jl[s];
                    taps[s] = issue_taps(bcast_f(P.tlx, v), bcast_f(P.tly, v), bcast_f(P.dxx, v), bcast_f(P.dxy, v),
                                         bcast_f(P.dyx, v), bcast_f(P.dyy, v), bcast_p(P.pix, v), bcast_i(P.W, v), ixx, iyy);
                }
            }
#pragma unroll
            for (int s = 0; s < S; s++) {
                rr[s] = gg[s] = bb[s] = 0.0f;
                if (s < ns) {
                    texel(taps[s], rr[s], gg[s], bb[s]);
                    if (active) {
                        float* bp = &L.buf[s * kTexStride];
                        bp[lane] = rr[s]; bp[kPlane + lane] = gg[s]; bp[2 * kPlane + lane] = bb[s];
                    }
                }
            }
            wave_lds_sync();
            PH_T(t4);
            PH_ADD(t3, t4, 14);
            // ---- 2. channel means
            {
                const int m = lane / 3, ch = lane - 3 * m;
                if (m < ns) L.ave[m * 4 + ch] = div_by(seq_sum<49>(&L.buf[m * kTexStride + ch * kPlane]), 1.0 / 49.0);
            }
            wave_lds_sync();
            // ---- 3. variance
#pragma unroll
            for (int s = 0; s < S; s++) {
                if (s < ns && active) {
                    const float4 av = *reinterpret_cast<const float4*>(&L.ave[s * 4]);
                    const float f0 = av.x - rr[s], f1 = av.y - gg[s], f2 = av.z - bb[s];
                    L.buf[s * kTexStride + lane] = f0 * f0 + f1 * f1 + f2 * f2;
                }
            }
            wave_lds_sync();
            if (lane < ns) {
                float v2 = sqrtf(div_by(seq_sum<49>(&L.buf[lane * kTexStride]), 1.0 / 147.0));
                if (v2 == 0.0f) v2 = 1.0f;
                L.inv[lane] = recip64(v2);
            }
            wave_lds_sync();
            PH_T(t5);
            PH_ADD(t4, t5, 15);
            // ---- 4. normalise; products with the request's reference texel
#pragma unroll
            for (int s = 0; s < S; s++) {
                if (s < ns) {
                    const float4 av = *reinterpret_cast<const float4*>(&L.ave[s * 4]);
                    float r = rr[s], g = gg[s], b = bb[s];
                    const double iv = L.inv[s];
                    r -= av.x; r = div_by(r, iv);
                    g -= av.y; g = div_by(g, iv);
                    b -= av.z; b = div_by(b, iv);
                    if ((refslots >> s) & 1u) { refn0 = r; refn1 = g; refn2 = b; }
                    else if (active) {
                        float* bp = &L.buf[s * kTexStride + 3 * lane];
                        bp[0] = refn0 * r; bp[1] = refn1 * g; bp[2] = refn2 * b;
                    }
                }
            }
            wave_lds_sync();
            PH_T(t6);
            PH_ADD(t5, t6, 20);
            // ---- 5. NCC sums
            float raw = 2.0f;
            if (lane < ns && !((refslots >> lane) & 1u)) raw = 1.0f - div_by(seq_sum<147>(&L.buf[lane * kTexStride]), 1.0 / 147.0);
            const float rob = robustincc(raw);  // lane-parallel: one division for the whole chunk
            PH_T(t7);
            PH_ADD(t6, t7, 21);
            // ---- deliver in sequence order
#pragma unroll
            for (int s = 0; s < S; s++) {
                if (s >= ns) break;
                const int v = jl[s];
                if ((refslots >> s) & 1u) {
                    if (cur >= 0) finalize();
                    cur = bcast_i(myp, v);
                    cur_obj = bcast_i(q.obj_mode, cur);
                    cur_rob = bcast_i(q.robust, cur);
                    const int refi = bcast_i(myi, v);
                    val = 0.0; nok = 0; ngr = 1;
                    mine = (lane == refi) ? 0.0f : 2.0f;
                } else {
                    ngr++;
                    if (cur_obj) { val += (double)bcast_f(rob, s); nok++; }
                    else if (lane == bcast_i(myi, v)) mine = cur_rob ? bcast_f(rob, s) : bcast_f(raw, s);
                }
            }
            wave_lds_sync();
            PH_T(t8);
            PH_ADD(t7, t8, 22);
        }
        if (cur >= 0) finalize();
        if (d_nok >= 0) out.obj = (d_nok < o.MIN_IMAGES_PER_PATCH - 1) ? 2.0 : d_val / (double)d_nok;
        // ---- requests whose reference image failed the gates (sampleTexture false for image `ref`)
        for (unsigned long long t = batch & ~refok; t; t &= t - 1ull) {
            const int p = __ffsll((long long)t) - 1;
            const int np = bcast_i(q.nimg, p);
            if (lane == p) {
                if (q.obj_mode) out.obj = 2.0;
                else {
                    float s = 0.0f;
                    for (int k = 0; k < np; k++) s += 2.0f;
                    out.keep = 0ull;
                    out.incc_sum = s;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// LANE-PER-GRAB exact-order evaluator: every (request, image) pair is one lane.
//
// The packed evaluator above spreads ONE grab over 49 lanes and has to hand the reference's sequential float32
// sums to single lanes through LDS (5 barriers and ~2.1 k wave-cycles per grab).  Here a lane does a whole
// sampleTexture by itself, exactly like the CPU thread of the reference does: the 49 bilinear taps in the
// reference's incremental stepping order (PatchOptimizer.cpp:511-526), the 147 raw values in registers,
// Patch2d::normalize's three sequential passes (Patch2d.hpp:46-84) and the 147-term dot product
// (Patch2d.hpp:37-44) with the request's reference texture, whose normalised values are read from the lane that
// grabbed the reference image with ds_bpermute (no LDS allocation, no barriers).  Up to 64 grabs advance per
// pass, the sums need no staging, and every float operation is the reference's in the reference's order.
// Requests are packed into passes like in wave_eval_packed (whole requests, <= 64 jobs per pass).
__device__ __forceinline__ void grab_window(const Prep& P, float (&t)[3 * kTexPix]) {
    float lx = P.tlx, ly = P.tly;  // row start: l += dy per row, c += dx per column
#pragma unroll
    for (int yy = 0; yy < kTex; yy++) {
        float cx = lx, cy = ly;
#pragma unroll
        for (int xx = 0; xx < kTex; xx++) {
            const int ix = (int)cx, iy = (int)cy;
            const uint8_t* p0 = P.pix + 3 * (iy * P.W + ix);
            Taps tp;
            tp.a = load8(p0);
            tp.b = load8(p0 + 3 * P.W);
            const float dx1 = cx - (float)ix, dx0 = 1.0f - dx1;
            const float dy1 = cy - (float)iy, dy0 = 1.0f - dy1;
            tp.f00 = dx0 * dy0; tp.f01 = dx0 * dy1; tp.f10 = dx1 * dy0; tp.f11 = dx1 * dy1;
            const int k = 3 * (yy * kTex + xx);
            texel(tp, t[k], t[k + 1], t[k + 2]);
            cx += P.dxx; cy += P.dxy;
        }
        lx += P.dyx; ly += P.dyy;
        // one row of taps (14 loads) in flight at a time: without this fence the scheduler hoists all 98 loads
        // of the window and spills the texture to scratch
        asm volatile("" ::: "memory");
    }
}

// Patch2d::normalize (Patch2d.hpp:46-84) on a register-resident texture
__device__ __forceinline__ void normalize_window(float (&t)[3 * kTexPix]) {
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int i = 0; i < kTexPix; i++) { s0 += t[3 * i]; s1 += t[3 * i + 1]; s2 += t[3 * i + 2]; }
    const float a0 = div_by(s0, 1.0 / 49.0), a1 = div_by(s1, 1.0 / 49.0), a2 = div_by(s2, 1.0 / 49.0);
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < kTexPix; i++) {
        const float f0 = a0 - t[3 * i], f1 = a1 - t[3 * i + 1], f2 = a2 - t[3 * i + 2];
        q += f0 * f0 + f1 * f1 + f2 * f2;
    }
    float v2 = sqrtf(div_by(q, 1.0 / 147.0));
    if (v2 == 0.0f) v2 = 1.0f;
    const double iv = recip64(v2);
#pragma unroll
    for (int i = 0; i < kTexPix; i++) {
        t[3 * i] = div_by(t[3 * i] - a0, iv);
        t[3 * i + 1] = div_by(t[3 * i + 1] - a1, iv);
        t[3 * i + 2] = div_by(t[3 * i + 2] - a2, iv);
    }
}

// myrow: the calling lane's own image list (uint16 ids; only read for lanes with a pending request).
// Must be called by all 64 lanes (wave-uniform control flow around every cross-lane operation).
#ifndef HPMVS_EVAL_JOBS_INLINE
#define HPMVS_EVAL_JOBS_ATTR __noinline__
#else
#define HPMVS_EVAL_JOBS_ATTR __forceinline__
#endif
// (out of line by default: its ~200 VGPRs -- the 147 texture values -- then do not compete with the caller's
//  long-lived per-patch registers inside the tap loop; the arguments travel by value, in registers)
template <bool COHERENT>
__device__ HPMVS_EVAL_JOBS_ATTR PackedRes wave_eval_jobs(DevScene sc, DevOptions o, const uint16_t* myrow,
                                                       unsigned long long pending, PackedReq q, PackedRes out) {
    const int lane = lane_id();
    const int q_refsel = q.obj_mode ? 0 : q.ref;
    const unsigned long long rowbits = (unsigned long long)myrow;
    const int row_lo = (int)(rowbits & 0xffffffffull), row_hi = (int)(rowbits >> 32);

    unsigned long long rem = pending;
    while (rem) {
        // ---- a pass: whole requests whose jobs fit the 64 lanes (at least one request)
        unsigned long long batch = 0ull;
        int J = 0;
        for (unsigned long long t = rem; t; t &= t - 1ull) {
            const int p = __ffsll((long long)t) - 1;
            const int np = bcast_i(q.nimg, p);
            if (batch && J + np > 64) break;
            batch |= 1ull << p;
            J += np;
        }
        rem &= ~batch;
        PH_T(t1);
        // ---- job of this lane (request myp, image index myi); each requesting lane learns where its jobs sit
        int myp = -1, myi = 0, refl = 0, myoff = 0, maxnp = 0;
        {
            int off = 0;
            for (unsigned long long t = batch; t; t &= t - 1ull) {
                const int p = __ffsll((long long)t) - 1;
                const int np = bcast_i(q.nimg, p);
                const int refp = bcast_i(q_refsel, p);
                if (lane >= off && lane < off + np) { myp = p; myi = lane - off; refl = off + refp; }
                if (lane == p) myoff = off;
                maxnp = np > maxnp ? np : maxnp;
                off += np;
            }
        }
        const int src = myp >= 0 ? myp : 0;
        float c[3], zax[3], pxa[3], pya[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            c[k] = shfl_f(q.c[k], src); zax[k] = shfl_f(q.zax[k], src);
            pxa[k] = shfl_f(q.px[k], src); pya[k] = shfl_f(q.py[k], src);
        }
        const float scale = shfl_f(q.scale, src);
        const unsigned long long rb = ((unsigned long long)(unsigned int)shfl_i(row_hi, src) << 32) |
                                      (unsigned long long)(unsigned int)shfl_i(row_lo, src);
        Prep P;
        P.ok = 0; P.tlx = P.tly = P.dxx = P.dxy = P.dyx = P.dyy = 0.0f; P.pix = nullptr; P.W = 0;
        if (myp >= 0) P = prep_image(sc.views[list_id<COHERENT>((const uint16_t*)rb, myi)], o, c, scale, pxa, pya, zax);
        const unsigned long long okmask = __ballot(P.ok != 0);
        PH_T(t2);
        PH_ADD(t1, t2, 13);

        // ---- the grab: sampleTexture + normalize, one lane per (request, image) whose reference image is usable
        const bool refok = (okmask >> refl) & 1ull;
        const bool work = myp >= 0 && P.ok && refok;
        float t[3 * kTexPix];
#pragma unroll
        for (int k = 0; k < 3 * kTexPix; k++) t[k] = 0.0f;
        if (work) {
            grab_window(P, t);
            normalize_window(t);
        }
        PH_T(t3);
        PH_ADD(t2, t3, 14);
        // ---- Patch2d::dot with the reference texture (held by lane refl)
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 3 * kTexPix; k++) acc += shfl_f(t[k], refl) * t[k];
        const float raw = 1.0f - div_by(acc, 1.0 / 147.0);
        const float rob = robustincc(raw);
        PH_T(t4);
        PH_ADD(t3, t4, 21);

        // ---- delivery: every requesting lane walks its own jobs in image order
        const bool mine = (batch >> lane) & 1ull;
        const int np = mine ? q.nimg : 0;
        const bool myrefok = mine && np > 0 && ((okmask >> (myoff + q_refsel)) & 1ull);
        double val = 0.0;
        float isum = 0.0f;
        unsigned long long keep = 0ull;
        int nok = 0, ngr = myrefok ? 1 : 0;
        for (int i = 0; i < maxnp; i++) {
            const int jl = (myoff + i) & 63;
            const float e_raw = shfl_f(raw, jl), e_rob = shfl_f(rob, jl);
            if (i < np) {
                const bool jok = myrefok && ((okmask >> jl) & 1ull);
                if (q.obj_mode) {
                    if (jok && i != q_refsel) { val += (double)e_rob; nok++; ngr++; }
                } else {
                    float e = 2.0f;
                    if (myrefok && i == q_refsel) e = 0.0f;
                    else if (jok) { e = q.robust ? e_rob : e_raw; ngr++; }
                    isum += e;
                    if (e < 1.0f - q.thr) keep |= 1ull << i;
                }
            }
        }
        if (mine) {
            out.ngrabs += ngr;
            if (q.obj_mode) out.obj = (!myrefok || nok < o.MIN_IMAGES_PER_PATCH - 1) ? 2.0 : val / (double)nok;
            else { out.keep = myrefok ? keep : 0ull; out.incc_sum = isum; }
        }
        PH_T(t5);
        PH_ADD(t4, t5, 22);
    }
    return out;
}

// ------------------------------------------------------------------------------------------
// Lane-per-grab evaluator with COMPACTION, for wavefronts that hold many requests (the stream kernel's EVAL visit:
// 64 requests of 64 different patches).  wave_eval_jobs gives a lane to every attached image, usable or not; with
// the long image lists after addImages most of them fail sampleTexture's gates (angle, level, window inside the
// image) and their lanes idle through the 5 k-instruction grab.  Here
//   phase 1  one lane per (request, image) runs ONLY the gates (prep_image, ~300 instructions): every request gets
//            the bit mask of its usable images;
//   phase 2  whole requests are packed by their USABLE image count, <= 64 grabs per pass, every lane of a pass
//            does a real grab (gates recomputed for the chosen image: cheaper than carrying 8 registers per job
//            through LDS); reference textures travel by ds_bpermute inside the pass as in wave_eval_jobs.
// Delivery walks each request's images in ascending index, unusable ones contributing the reference's constants
// (2.0 for a failed grab, PatchOptimizer.cpp:458-470), so every sum is formed in the reference's order.
#ifdef SK_STATS
__device__ unsigned long long* g_evdbg = nullptr;  // 16 counters, set by the stream kernel (stats builds only)
#define EV_T(v) const long long v = clock64()
#define EV_ACC(i, a, b) do { if (lane_id() == 0 && g_evdbg) atomicAdd(g_evdbg + (i), (unsigned long long)((b) - (a))); } while (0)
#else
#define EV_T(v) do { } while (0)
#define EV_ACC(i, a, b) do { } while (0)
#endif
template <bool COHERENT>
__device__ __noinline__ PackedRes wave_eval_compact(DevScene sc, DevOptions o, const uint16_t* myrow,
                                                    unsigned long long pending, PackedReq q, PackedRes out) {
    const int lane = lane_id();
    const int q_refsel = q.obj_mode ? 0 : q.ref;
    const unsigned long long rowbits = (unsigned long long)myrow;
    const int row_lo = (int)(rowbits & 0xffffffffull), row_hi = (int)(rowbits >> 32);
    const bool has_req = (pending >> lane) & 1ull;

    // ---------------- phase 1: gates only
    unsigned long long mymask = 0ull;  // requesting lanes: bit i <=> image i of the list passes the gates
    EV_T(e0);
    for (unsigned long long rem = pending; rem;) {
        EV_ACC(1, 0, 1);
        unsigned long long batch = 0ull;
        int J = 0;
        for (unsigned long long t = rem; t; t &= t - 1ull) {
            const int p = __ffsll((long long)t) - 1;
            const int np = bcast_i(q.nimg, p);
            if (batch && J + np > 64) break;
            batch |= 1ull << p;
            J += np;
        }
        rem &= ~batch;
        int myp = -1, myi = 0, myoff = 0;
        {
            int off = 0;
            for (unsigned long long t = batch; t; t &= t - 1ull) {
                const int p = __ffsll((long long)t) - 1;
                const int np = bcast_i(q.nimg, p);
                if (lane >= off && lane < off + np) { myp = p; myi = lane - off; }
                if (lane == p) myoff = off;
                off += np;
            }
        }
        const int src = myp >= 0 ? myp : 0;
        float c[3], zax[3], pxa[3], pya[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            c[k] = shfl_f(q.c[k], src); zax[k] = shfl_f(q.zax[k], src);
            pxa[k] = shfl_f(q.px[k], src); pya[k] = shfl_f(q.py[k], src);
        }
        const float scale = shfl_f(q.scale, src);
        const unsigned long long rb = ((unsigned long long)(unsigned int)shfl_i(row_hi, src) << 32) |
                                      (unsigned long long)(unsigned int)shfl_i(row_lo, src);
        int okj = 0;
        if (myp >= 0) okj = prep_image(sc.views[list_id<COHERENT>((const uint16_t*)rb, myi)], o, c, scale, pxa, pya, zax).ok;
        const unsigned long long okmask = __ballot(okj != 0);
        if ((batch >> lane) & 1ull) {
            const int np = q.nimg;
            const unsigned long long m = (np >= 64) ? ~0ull : ((1ull << np) - 1ull);
            mymask = (okmask >> myoff) & m;
        }
    }
    const bool refok = has_req && q.nimg > 0 && ((mymask >> q_refsel) & 1ull);
    if (has_req && !refok) {  // sampleTexture failed for the reference image (PatchOptimizer.cpp:291-293, 454-457)
        if (q.obj_mode) out.obj = 2.0;
        else {
            float s2 = 0.0f;
            for (int k = 0; k < q.nimg; k++) s2 += 2.0f;
            out.keep = 0ull;
            out.incc_sum = s2;
        }
    }
    const int mycnt = refok ? __popcll(mymask) : 0;  // grabs of this request, reference image included
    EV_T(e1);
    EV_ACC(0, e0, e1);

    // ---------------- phase 2: grabs
    for (unsigned long long rem = __ballot(refok); rem;) {
        EV_T(p0);
        unsigned long long batch = 0ull;
        int J = 0, maxnp = 0;
        for (unsigned long long t = rem; t; t &= t - 1ull) {
            const int p = __ffsll((long long)t) - 1;
            const int cp = bcast_i(mycnt, p);
            if (batch && J + cp > 64) break;
            batch |= 1ull << p;
            J += cp;
            const int np = bcast_i(q.nimg, p);
            maxnp = np > maxnp ? np : maxnp;
        }
        rem &= ~batch;
        // ---- lane -> (request myp, j-th usable image); every requesting lane learns where its grabs sit
        int myp = -1, myj = 0, myoff = 0;
        {
            int off = 0;
            for (unsigned long long t = batch; t; t &= t - 1ull) {
                const int p = __ffsll((long long)t) - 1;
                const int cp = bcast_i(mycnt, p);
                if (lane >= off && lane < off + cp) { myp = p; myj = lane - off; }
                if (lane == p) myoff = off;
                off += cp;
            }
        }
        const int src = myp >= 0 ? myp : 0;
        const unsigned long long pmask = ((unsigned long long)(unsigned int)shfl_i((int)(mymask >> 32), src) << 32) |
                                         (unsigned long long)(unsigned int)shfl_i((int)(mymask & 0xffffffffull), src);
        const int prefsel = shfl_i(q_refsel, src);
        const int poff = shfl_i(myoff, src);
        int myi = 0, refl = 0;
        if (myp >= 0) {
            unsigned long long m = pmask;
            for (int t = 0; t < myj; t++) m &= m - 1ull;  // drop the myj lowest usable images
            myi = __ffsll((long long)m) - 1;
            refl = poff + __popcll(pmask & ((1ull << prefsel) - 1ull));
        }
        float c[3], zax[3], pxa[3], pya[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            c[k] = shfl_f(q.c[k], src); zax[k] = shfl_f(q.zax[k], src);
            pxa[k] = shfl_f(q.px[k], src); pya[k] = shfl_f(q.py[k], src);
        }
        const float scale = shfl_f(q.scale, src);
        const unsigned long long rb = ((unsigned long long)(unsigned int)shfl_i(row_hi, src) << 32) |
                                      (unsigned long long)(unsigned int)shfl_i(row_lo, src);
        float t[3 * kTexPix];
#pragma unroll
        for (int k = 0; k < 3 * kTexPix; k++) t[k] = 0.0f;
        EV_T(p1);
        Prep P;
        P.ok = 0; P.tlx = P.tly = P.dxx = P.dxy = P.dyx = P.dyy = 0.0f; P.pix = nullptr; P.W = 0;
        if (myp >= 0) P = prep_image(sc.views[list_id<COHERENT>((const uint16_t*)rb, myi)], o, c, scale, pxa, pya, zax);
        EV_T(p2);
        if (myp >= 0) grab_window(P, t);   // P.ok holds: same inputs, same operations as in phase 1
        EV_T(p3);
        if (myp >= 0) normalize_window(t);
        EV_T(p4);
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 3 * kTexPix; k++) acc += shfl_f(t[k], refl) * t[k];
        const float raw = 1.0f - div_by(acc, 1.0 / 147.0);
        const float rob = robustincc(raw);
        EV_T(p5);

        // ---- delivery: every requesting lane of the pass walks its image list in index order
        const bool mine = (batch >> lane) & 1ull;
        const int np = mine ? q.nimg : 0;
        double val = 0.0;
        float isum = 0.0f;
        unsigned long long keep = 0ull;
        int nok = 0;
        for (int i = 0; i < maxnp; i++) {
            const bool usable = mine && ((mymask >> i) & 1ull);
            const int jl = (myoff + __popcll(mymask & ((1ull << i) - 1ull))) & 63;
            const float e_raw = shfl_f(raw, jl), e_rob = shfl_f(rob, jl);
            if (i < np) {
                if (q.obj_mode) {
                    if (usable && i != q_refsel) { val += (double)e_rob; nok++; }
                } else {
                    float e = 2.0f;
                    if (i == q_refsel) e = 0.0f;
                    else if (usable) e = q.robust ? e_rob : e_raw;
                    isum += e;
                    if (e < 1.0f - q.thr) keep |= 1ull << i;
                }
            }
        }
        if (mine) {
            out.ngrabs += mycnt;
            if (q.obj_mode) out.obj = (nok < o.MIN_IMAGES_PER_PATCH - 1) ? 2.0 : val / (double)nok;
            else { out.keep = keep; out.incc_sum = isum; }
        }
        EV_T(p6);
        EV_ACC(2, p0, p1); EV_ACC(3, p1, p2); EV_ACC(4, p2, p3); EV_ACC(5, p3, p4); EV_ACC(6, p4, p5); EV_ACC(7, p5, p6);
        EV_ACC(8, 0, 1); EV_ACC(9, 0, (long long)J);
    }
    return out;
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
