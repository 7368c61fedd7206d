// kernel_objective_lane.hip -- objective_fn for a batch with ONE LANE PER GRAB (gfx950).
//
// The refinement kernel's photometric service gives a grab seven lanes (photometric.hpp: one lane per window row), because
// there the optimiser's state owns the registers and the LDS.  The price is in the reference's sequential float32 sums
// (Patch2d::normalize / dot, Patch2d.hpp:37-84): a 49- or 147-term chain that runs through the seven lanes of a group is
// issued as 7 x 7 (or 7 x 21) wavefront instructions of which one row's lanes do useful work -- 294 add instructions for
// the nine grabs of a pass, a third of the service's vector instructions (profiles/r05/pmc_objective).
//
// A kernel that does nothing but evaluate objectives has the registers for the other layout: a lane samples the WHOLE 7 x 7 x 3
// window of its (patch, image) pair (147 float32 values in registers, sampleTexture's incremental stepping,
// PatchOptimizer.cpp:511-526), runs the three sums of Patch2d::normalize over its own values -- every lane of the
// wavefront a chain of its own, 64 grabs per chain instruction instead of nine -- and the lanes of a patch's other images
// fetch the normalised texels of the patch's reference image (lane of image 0) one ds_bpermute per product for
// Patch2d::dot.  Same float32 operations in the same order as the reference, so the values are the reference's bit for
// bit (tests/test_gpu_photometric.py compares this kernel with the row-lane evaluator and with the oracle).
//
// Work distribution: a wavefront owns a contiguous range of patches and takes, per pass, as many WHOLE patches as fit its
// 64 lanes (prefix sum over the image counts; a patch has at most kMaxImages = 64 images).  objective_fn's float64 sum
// over a patch's usable images (PatchOptimizer.cpp:299-309) is formed on the patch's first lane in index order.
#include <hip/hip_runtime.h>

#include "launch.h"
#include "photometric.hpp"

namespace hpmvs {

namespace {

constexpr int kWin = kTex * kTex;       // 49 texels
constexpr int kWinVals = 3 * kWin;      // 147 float32 values

__device__ __forceinline__ bool lane_list_ok(const DevScene& sc, const DevBatch& b, int p, int nimg) {
    if (!(nimg > 0 && nimg <= kMaxImages && nimg <= b.max_images)) return false;
    const int32_t* list = b.images + (size_t)p * b.max_images;
    bool bad = false;
    for (int k = 0; k < nimg; k++) { const int id = list[k]; bad |= (id < 0 || id >= sc.n_views); }
    return !bad;
}

}  // namespace

__global__ void __launch_bounds__(64, 2) objective_lane_kernel(DevScene sc, DevOptions o, DevBatch b, int per_wave,
                                                               const double* __restrict__ xs, double* __restrict__ f_out,
                                                               int32_t* __restrict__ ngrabs_out) {
    const int lane = lane_id();
    const int lo_p = blockIdx.x * per_wave;
    const int hi_p = (lo_p + per_wave < b.n) ? lo_p + per_wave : b.n;
    int cur = lo_p;
    while (cur < hi_p) {
        // ---- the patches of this pass: lanes 0.. hold the next patches; whole patches while their images fit 64 lanes
        const int p = cur + lane;
        const bool cand = p < hi_p;
        int nimg = 0;
        bool listok = false;
        if (cand) {
            nimg = b.n_images[p];
            listok = lane_list_ok(sc, b, p, nimg);
        }
        const int mynp = (cand && listok) ? nimg : 0;   // a patch with a bad list is answered with 2.0 and uses no lane
        const int incl = wave_scan_add(mynp);
        const int excl = incl - mynp;
        const unsigned long long fit = __ballot(cand && incl <= 64);   // (incl is non-decreasing: a prefix of the lanes)
        const int taken = __popcll(fit);                               // >= 1: one patch always fits
        const bool mine = lane < taken;
        const int J = bcast_i(incl, taken - 1);                        // job lanes in use
        cur += taken;

        // ---- the patch's frame at x (optimizePatch's fields, setCenterNorm, calculatePatchAxis), on the patch's lane
        float qc[3] = {0.0f, 0.0f, 0.0f}, qz[3] = {0.0f, 0.0f, 0.0f}, qx[3] = {0.0f, 0.0f, 0.0f}, qy[3] = {0.0f, 0.0f, 0.0f};
        float qscale = 1.0f;
        const int32_t* list = b.images;
        if (mine && listok) {
            list = b.images + (size_t)p * b.max_images;
            const float c0[3] = {b.center[4 * p], b.center[4 * p + 1], b.center[4 * p + 2]};
            OptFields F;
            set_optimization_fields(sc.views[list[0]], c0, F);
            const double x[3] = {xs[3 * p], xs[3 * p + 1], xs[3 * p + 2]};
            float n[3];
            set_center_norm(F, o, x, qc, n);
            calc_patch_axes(sc.views[list[0]].xaxis, sc.views[list[0]].yaxis, n, b.scale[p], qx, qy, qz);
            qscale = b.scale[p];
        }

        // ---- my job: lane j <-> (patch lane r, image k) with excl[r] <= j < incl[r]
        int r = 0;
#pragma unroll
        for (int step = 32; step >= 1; step >>= 1) {
            const int v = shfl_i(incl, r + step - 1);
            r += (v <= lane) ? step : 0;
        }
        const bool has = lane < J;
        r = has ? r : 0;
        const int e = shfl_i(excl, r);
        const int k = has ? lane - e : 0;
        float jc[3], jz[3], jx[3], jy[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            jc[a] = shfl_f(qc[a], r); jz[a] = shfl_f(qz[a], r); jx[a] = shfl_f(qx[a], r); jy[a] = shfl_f(qy[a], r);
        }
        const float jscale = shfl_f(qscale, r);
        const unsigned long long listbits = (unsigned long long)list;
        const unsigned long long lb = ((unsigned long long)(unsigned int)shfl_i((int)(listbits >> 32), r) << 32) |
                                      (unsigned long long)(unsigned int)shfl_i((int)(listbits & 0xffffffffull), r);
        int id = 0;
        if (has) id = ((const int32_t*)lb)[k];   // (a lane without a job prepares view 0 and drops the result)
        const ViewRegs VR = load_view(sc.views[id]);
        Prep P = prep_from_view(sc, VR, o, jc, jscale, jx, jy, jz);
        const unsigned long long okmask = __ballot(has && P.ok != 0);
        // sampleTexture false for the reference image: objective_fn returns 2.0 before it looks at the others
        const bool refok = (okmask >> e) & 1ull;
        const bool act = has && P.ok != 0 && refok;

        // ---- the window: sampleTexture's stepping (l += dy per row, c += dx per column), Image::getColor per texel
        float t[kWinVals];
        {
            float rx = P.tlx, ry = P.tly;
            // (no branch around the loads -- the compiler would merge the guarded blocks of all rows into one and keep 98 tap
            //  pairs alive: a lane without a grab reads the first bytes of a view's level 0 and drops them)
            const uint8_t* pix = act ? P.pix : VR.pix0;
            const int W = act ? P.W : 0;
            const int amask = act ? -1 : 0;
#pragma unroll
            for (int yy = 0; yy < kTex; yy++) {
                float x = rx, y = ry;
                uint2 ta[kTex], tb[kTex];
                float wx[kTex], wy[kTex];
#pragma unroll
                for (int xx = 0; xx < kTex; xx++) {
                    const int lx = (int)x, ly = (int)y;
                    wx[xx] = x - (float)lx; wy[xx] = y - (float)ly;
                    const int toff = (3 * (ly * W + lx)) & amask;
                    const uint8_t* p0 = pix + toff;
                    ta[xx] = load8(p0);
                    tb[xx] = load8(p0 + 3 * W);
                    x += P.dxx; y += P.dxy;
                }
                float* row = &t[3 * kTex * yy];
#pragma unroll
                for (int xx = 0; xx + 1 < kTex; xx += 2)
                    texel_pair(ta[xx], tb[xx], wx[xx], wy[xx], ta[xx + 1], tb[xx + 1], wx[xx + 1], wy[xx + 1], &row[3 * xx], &row[3 * xx + 3]);
                {
                    const int xx = kTex - 1;
                    Taps tp;
                    tp.a = ta[xx]; tp.b = tb[xx];
                    const float dx1 = wx[xx], dx0 = 1.0f - dx1;
                    const float dy1 = wy[xx], dy0 = 1.0f - dy1;
                    tp.f00 = dx0 * dy0; tp.f01 = dx0 * dy1; tp.f10 = dx1 * dy0; tp.f11 = dx1 * dy1;
                    texel(tp, row[3 * xx], row[3 * xx + 1], row[3 * xx + 2]);
                }
                rx += P.dyx; ry += P.dyy;
                // a row's taps are consumed before the next row's are fetched (registers: the compiler would otherwise issue
                // the loads of all rows first and sink the arithmetic below them)
#pragma unroll
                for (int i = 0; i < 3 * kTex; i += 3) asm volatile("" : "+v"(row[i]), "+v"(row[i + 1]), "+v"(row[i + 2]) :: "memory");
            }
        }

        // ---- Patch2d::normalize (Patch2d.hpp:46-84): channel means, variance, normalisation -- sequential sums in pixel order
        float a0, a1, a2;
        {
            float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
#pragma unroll
            for (int i = 0; i < kWin; i++) { s0 += t[3 * i]; s1 += t[3 * i + 1]; s2 += t[3 * i + 2]; }
            a0 = div_by(s0, 1.0 / 49.0); a1 = div_by(s1, 1.0 / 49.0); a2 = div_by(s2, 1.0 / 49.0);
        }
        // t <- t - mean.  Patch2d::normalize squares (mean - t) for the variance and divides (t - mean): the one is the exact
        // negative of the other and x * x == (-x) * (-x) bit for bit, so one subtraction serves both.
#pragma unroll
        for (int i = 0; i < kWin; i++) { t[3 * i] -= a0; t[3 * i + 1] -= a1; t[3 * i + 2] -= a2; }
        double iv;
        {
            float sv = 0.0f;
#pragma unroll
            for (int i = 0; i < kWin; i++) sv += t[3 * i] * t[3 * i] + t[3 * i + 1] * t[3 * i + 1] + t[3 * i + 2] * t[3 * i + 2];
            float v2 = sqrtf(div_by(sv, 1.0 / 147.0));
            if (v2 == 0.0f) v2 = 1.0f;
            iv = recip64(v2);
        }
#pragma unroll
        for (int i = 0; i < kWinVals; i++) {
            t[i] = div_by(t[i], iv);
            if (i % 21 == 20) __builtin_amdgcn_sched_barrier(0);
        }

        // ---- Patch2d::dot with the patch's reference texture (Patch2d.hpp:37-44): the lane of image 0 holds it
        // (the fetch address passes through an opaque asm that also takes the running sum of two rows earlier: at most two
        //  rows of fetched texels are in flight -- the compiler would otherwise issue all 147 fetches first and spill)
        int refaddr = (has ? e : lane) << 2;
        float dot = 0.0f, dlag = 0.0f;
#pragma unroll
        for (int yy = 0; yy < kTex; yy++) {
            asm volatile("" : "+v"(refaddr) : "v"(dlag));
            dlag = dot;
#pragma unroll
            for (int i = 3 * kTex * yy; i < 3 * kTex * (yy + 1); i++)
                dot += __int_as_float(__builtin_amdgcn_ds_bpermute(refaddr, __float_as_int(t[i]))) * t[i];
        }
        const float raw = 1.0f - div_by(dot, 1.0 / 147.0);
        const float rob = robustincc(raw);

        // ---- objective_fn (PatchOptimizer.cpp:299-309): float64 sum over the usable images other than the reference, in
        //      index order, on the patch's lane
        {
            const unsigned long long myok = mynp > 0 ? ((okmask >> excl) & (mynp >= 64 ? ~0ull : ((1ull << mynp) - 1ull))) : 0ull;
            const bool fin = mine && mynp > 0 && (myok & 1ull);
            const int maxn = wave_max_i32(mine ? mynp : 0);
            double val = 0.0;
            int nok = 0;
            for (int kk = 1; kk < maxn; kk++) {
                const float v = shfl_f(rob, (excl + kk) & 63);
                const bool add = fin && kk < mynp && ((myok >> kk) & 1ull);
                val += add ? (double)v : 0.0;
                nok += add ? 1 : 0;
            }
            if (mine) {
                f_out[p] = fin ? ((nok < o.MIN_IMAGES_PER_PATCH - 1) ? 2.0 : val / (double)nok) : 2.0;
                if (ngrabs_out) ngrabs_out[p] = fin ? __popcll(myok) : 0;
            }
        }
    }
}

void launch_objective_lane(const DevScene& sc, const DevOptions& o, const DevBatch& b, const double* xs, double* f_out,
                           int32_t* ngrabs_out, hipStream_t st) {
    if (b.n <= 0) return;
    // contiguous patch ranges, one per wavefront: enough wavefronts to fill the chip eight deep, at least ~16 passes each
    // (a range's last pass is partly empty), never fewer than 64 patches
    const int max_waves = 256 * 8;
    int per_wave = (b.n + max_waves - 1) / max_waves;
    if (per_wave < 64) per_wave = 64;
    const int blocks = (b.n + per_wave - 1) / per_wave;
    hipLaunchKernelGGL(objective_lane_kernel, dim3(blocks), dim3(64), 0, st, sc, o, b, per_wave, xs, f_out, ngrabs_out);
}

}  // namespace hpmvs
