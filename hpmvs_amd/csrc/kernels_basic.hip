// kernels_basic.hip -- gfx950 kernels: pyramid build, batched objective, batched setINCCs.
// (The full refinement kernel lives in kernel_optimize.hip.)
#include <hip/hip_runtime.h>

#include "launch.h"
#include "photometric.hpp"

namespace hpmvs {

// ------------------------------------------------------------------------------------------
// CImg get_resize_halfXY on interleaved u8 RGB (reference thirdLibs/cimg/CImg.h:21189-21203 with
// the clamped 3x3 walk of cimg_for3x3, CImg.h:1159-1175; called from Image::load,
// src/hpmvs/Image.cpp:55-57).  One thread per output pixel (3 channels), rows of threads read
// consecutive source pixels, so a wave covers 64*2*3 contiguous bytes of each of three rows.
// No FMA contraction: the u8 truncation must match the CPU bit for bit.
__global__ void __launch_bounds__(256) half_resize_kernel(const uint8_t* __restrict__ src, int w, int h,
                                                          uint8_t* __restrict__ dst) {
#pragma clang fp contract(off)
    const int w2 = w / 2, h2 = h / 2;
    const int ox = blockIdx.x * blockDim.x + threadIdx.x;
    const int oy = blockIdx.y;
    if (ox >= w2 || oy >= h2) return;
    const int x = 2 * ox + 1, y = 2 * oy + 1;
    const int xp = x - 1, xn = (x + 1 < w) ? x + 1 : w - 1;
    const int yp = y - 1, yn = (y + 1 < h) ? y + 1 : h - 1;
    const float m0 = 0.07842776544f, m1 = 0.1231940459f, m4 = 0.1935127547f;
    const uint8_t* r0 = src + (size_t)yp * w * 3;
    const uint8_t* r1 = src + (size_t)y * w * 3;
    const uint8_t* r2 = src + (size_t)yn * w * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float v = (float)r0[3 * xp + c] * m0;
        v = v + (float)r0[3 * x + c] * m1;
        v = v + (float)r0[3 * xn + c] * m0;
        v = v + (float)r1[3 * xp + c] * m1;
        v = v + (float)r1[3 * x + c] * m4;
        v = v + (float)r1[3 * xn + c] * m1;
        v = v + (float)r2[3 * xp + c] * m0;
        v = v + (float)r2[3 * x + c] * m1;
        v = v + (float)r2[3 * xn + c] * m0;
        dst[3 * ((size_t)oy * w2 + ox) + c] = (uint8_t)v;
    }
}

void launch_half_resize(const uint8_t* src, int w, int h, uint8_t* dst, hipStream_t st) {
    const int w2 = w / 2, h2 = h / 2;
    if (w2 <= 0 || h2 <= 0) return;
    dim3 block(256), grid((w2 + 255) / 256, h2);
    hipLaunchKernelGGL(half_resize_kernel, grid, block, 0, st, src, w, h, dst);
}

// ------------------------------------------------------------------------------------------
// objective_fn / setINCCs for a batch, through the product evaluator (wave_eval_rows): a wavefront takes kGroups
// patches at a time, lane r < kGroups posts the request of patch base + r -- the same request / job / group
// machinery the refinement kernel drives, so the photometric parity tests exercise the product code.

// image ids come straight from the caller: a list that is longer than its row, or an id outside the scene, is
// answered with the plateau value (2.0), never dereferenced
__device__ __forceinline__ bool list_ok(const DevScene& sc, const DevBatch& b, int p, int nimg) {
    if (!(nimg > 0 && nimg <= kMaxImages && nimg <= b.max_images)) return false;
    const int32_t* list = b.images + (size_t)p * b.max_images;
    bool bad = false;
    for (int k = 0; k < nimg; k++) { const int id = list[k]; bad |= (id < 0 || id >= sc.n_views); }
    return !bad;
}

// (HPMVS_OBJ_WAVES_PER_EU: measurement builds only -- what the photometric service costs under the register budget of 2 / 3 / 4
//  wavefronts per SIMD, profiles/r05/split_gate.txt)
#ifndef HPMVS_OBJ_WAVES_PER_EU
#define HPMVS_OBJ_WAVES_PER_EU 2
#endif
__global__ void __launch_bounds__(64, HPMVS_OBJ_WAVES_PER_EU) objective_kernel(DevScene sc, DevOptions o, DevBatch b,
                                                       const double* __restrict__ xs, double* __restrict__ f_out,
                                                       int32_t* __restrict__ ngrabs_out) {
    const int lane = lane_id();
    __shared__ float erow_s[kGroups * kMaxImages];  // the requests' result rows (wave_eval_rows)
    for (int base = blockIdx.x * kGroups; base < b.n; base += gridDim.x * kGroups) {
        const int p = base + lane;
        const bool mine = lane < kGroups && p < b.n;
        PackedReq q;
        q.obj_mode = 1; q.ref = 0; q.robust = 0; q.nimg = 0; q.thr = 0.0f; q.scale = 1.0f;
        for (int k = 0; k < 3; k++) { q.c[k] = 0.0f; q.zax[k] = 0.0f; q.px[k] = 0.0f; q.py[k] = 0.0f; }
        const int32_t* list = b.images;
        bool ok = false;
        if (mine) {
            const int nimg = b.n_images[p];
            ok = list_ok(sc, b, p, nimg);
            if (ok) {
                list = b.images + (size_t)p * b.max_images;
                const float c0[3] = {b.center[4 * p], b.center[4 * p + 1], b.center[4 * p + 2]};
                OptFields F;
                set_optimization_fields(sc.views[list[0]], c0, F);
                const double x[3] = {xs[3 * p], xs[3 * p + 1], xs[3 * p + 2]};
                float c[3], n[3], px[3], py[3], pz[3];
                set_center_norm(F, o, x, c, n);
                calc_patch_axes(sc.views[list[0]].xaxis, sc.views[list[0]].yaxis, n, b.scale[p], px, py, pz);
                q.nimg = nimg; q.scale = b.scale[p];
                for (int k = 0; k < 3; k++) { q.c[k] = c[k]; q.zax[k] = pz[k]; q.px[k] = px[k]; q.py[k] = py[k]; }
            }
        }
        PackedRes r;
        r.obj = 2.0; r.keep = 0ull; r.incc_sum = 0.0f; r.ngrabs = 0;
        wave_eval_rows<false>(sc, o, list, __ballot(mine && ok), q, r, LdsRows{erow_s, kMaxImages});
        if (mine) {
            f_out[p] = ok ? r.obj : 2.0;
            if (ngrabs_out) ngrabs_out[p] = ok ? r.ngrabs : 0;
        }
    }
}

void launch_objective(const DevScene& sc, const DevOptions& o, const DevBatch& b, const double* xs, double* f_out,
                      int32_t* ngrabs_out, hipStream_t st) {
    if (b.n <= 0) return;
    int blocks = (b.n + kGroups - 1) / kGroups;
    if (blocks > 256 * 4 * HPMVS_OBJ_WAVES_PER_EU) blocks = 256 * 4 * HPMVS_OBJ_WAVES_PER_EU;
    hipLaunchKernelGGL(objective_kernel, dim3(blocks), dim3(64), 0, st, sc, o, b, xs, f_out, ngrabs_out);
}

// setINCCs(ref_idx, robust) for a batch: out[p][k] = inccs[k] for k < n_images, 0 beyond
__global__ void __launch_bounds__(64) inccs_kernel(DevScene sc, DevOptions o, DevBatch b, int ref_idx, int robust,
                                                   float* __restrict__ out) {
    const int lane = lane_id();
    for (int base = blockIdx.x * kGroups; base < b.n; base += gridDim.x * kGroups) {
        const int p = base + lane;
        const bool mine = lane < kGroups && p < b.n;
        PackedReq q;
        q.obj_mode = 0; q.ref = ref_idx; q.robust = robust; q.nimg = 0; q.thr = 0.0f; q.scale = 1.0f;
        for (int k = 0; k < 3; k++) { q.c[k] = 0.0f; q.zax[k] = 0.0f; q.px[k] = 0.0f; q.py[k] = 0.0f; }
        const int32_t* list = b.images;
        float* row = out;
        int nimg = 0;
        bool ok = false;
        if (mine) {
            nimg = b.n_images[p];
            row = out + (size_t)p * b.max_images;
            // same validation as objective_kernel: bad lists give inccs = 2.0 for every image
            ok = list_ok(sc, b, p, nimg) && ref_idx >= 0 && ref_idx < nimg;
            if (ok) {
                list = b.images + (size_t)p * b.max_images;
                const float n[3] = {b.normal[4 * p], b.normal[4 * p + 1], b.normal[4 * p + 2]};
                float px[3], py[3], pz[3];
                calc_patch_axes(sc.views[list[ref_idx]].xaxis, sc.views[list[ref_idx]].yaxis, n, b.scale[p], px, py, pz);
                q.nimg = nimg; q.scale = b.scale[p];
                for (int k = 0; k < 3; k++) { q.c[k] = b.center[4 * p + k]; q.zax[k] = n[k]; q.px[k] = px[k]; q.py[k] = py[k]; }
            }
        }
        PackedRes r;
        r.obj = 2.0; r.keep = 0ull; r.incc_sum = 0.0f; r.ngrabs = 0;
        // the requests' result rows are the output rows themselves: requesting lane r <-> patch base + r
        wave_eval_rows<true>(sc, o, list, __ballot(mine && ok), q, r, GlobalRows{out + (size_t)base * b.max_images, b.max_images});
        if (mine) {
            // the evaluator stored the grabbed images' values and the reference image's 0 (r.keep = grabbed mask);
            // the rest is 2.0
            const bool refok = ok && r.ngrabs > 0;
            for (int k = 0; k < b.max_images; k++) {
                const bool grabbed = refok && ((r.keep >> k) & 1ull);
                if (k >= nimg) row[k] = 0.0f;
                else if (!grabbed) row[k] = 2.0f;
            }
        }
    }
}

void launch_inccs(const DevScene& sc, const DevOptions& o, const DevBatch& b, int ref_idx, int robust, float* out,
                  hipStream_t st) {
    if (b.n <= 0) return;
    int blocks = (b.n + kGroups - 1) / kGroups;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(inccs_kernel, dim3(blocks), dim3(64), 0, st, sc, o, b, ref_idx, robust, out);
}

// ------------------------------------------------------------------------------------------
// Seed construction of Scene::initPatches (reference src/hpmvs/Scene.cpp:118-163), one thread per
// NVM point.  Seeds that the reference skips before optimize() get n_images = -reason; the
// refinement kernel drops them immediately with stage = reason.
__global__ void __launch_bounds__(256) seed_init_kernel(DevScene sc, DevOptions o, int start_level, int n,
                                                        const double* __restrict__ xyz,
                                                        const int32_t* __restrict__ meas_off,
                                                        const int32_t* __restrict__ meas_img, DevBatch b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float c[3] = {(float)xyz[3 * i], (float)xyz[3 * i + 1], (float)xyz[3 * i + 2]};
    int32_t* row = b.images + (size_t)i * b.max_images;
    int nimg = 0, why = 0;
    const int m0 = meas_off[i], m1 = meas_off[i + 1];
    if (m1 - m0 < o.MIN_IMAGES_PER_PATCH) why = 10;
    else {
        const float s = __int_as_float((127 - start_level) << 23);
        const int margin = 2;  // cSize, Scene.cpp:91,143
        for (int m = m0; m < m1; m++) {
            const int idx = meas_img[m];
            if (idx < 0 || idx >= sc.n_views) continue;
            const DevView& V = sc.views[idx];
            float P[12];
            for (int k = 0; k < 12; k++) P[k] = V.P0[k];
            float u, v;
            project0(P, c[0], c[1], c[2], u, v);
            u *= s; v *= s;
            if (u < (float)margin || v < (float)margin || u >= (float)(V.w[start_level] - margin) ||
                v >= (float)(V.h[start_level] - margin))
                continue;
            if (nimg < b.max_images) row[nimg] = idx;
            nimg++;
        }
        if (nimg < 2) why = 11;
        else if (nimg > b.max_images) why = 100;
    }
    float nrm[3] = {0.0f, 0.0f, 0.0f};
    float scale = 0.0f;
    if (!why) {
        const DevView& V0 = sc.views[row[0]];
        const float d[3] = {V0.center[0] - c[0], V0.center[1] - c[1], V0.center[2] - c[2]};
        normalized3f(d, nrm);  // Vector4f::normalize() with w = 1 - 1 = 0
        const float e[3] = {c[0] - V0.center[0], c[1] - V0.center[1], c[2] - V0.center[2]};
        const float fz = sqrtf(dot3f(e, e));
        scale = (V0.fsum == 0.0f) ? 1.0f
                                  : (float)(2.0 * (double)fz * (double)(1 << start_level) / (double)V0.fsum);  // getScale
    }
    b.center[4 * i] = c[0]; b.center[4 * i + 1] = c[1]; b.center[4 * i + 2] = c[2]; b.center[4 * i + 3] = 1.0f;
    b.normal[4 * i] = nrm[0]; b.normal[4 * i + 1] = nrm[1]; b.normal[4 * i + 2] = nrm[2]; b.normal[4 * i + 3] = 0.0f;
    b.scale[i] = scale;
    b.n_images[i] = why ? -why : nimg;  // negative: rejected before optimize(); the refinement kernel reports -n as stage
    b.ok[i] = 0;
}

void launch_seed_init(const DevScene& sc, const DevOptions& o, int start_level, int n, const double* xyz,
                      const int32_t* meas_off, const int32_t* meas_img, const DevBatch& b, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(seed_init_kernel, dim3((n + 255) / 256), dim3(256), 0, st, sc, o, start_level, n, xyz, meas_off,
                       meas_img, b);
}

// |centre - xyz| > 2 * scale -> drop (Scene.cpp:171)
__global__ void __launch_bounds__(256) drift_gate_kernel(int n, const double* __restrict__ xyz, DevBatch b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (b.ok[i]) {
        const float d[3] = {b.center[4 * i] - (float)xyz[3 * i], b.center[4 * i + 1] - (float)xyz[3 * i + 1],
                            b.center[4 * i + 2] - (float)xyz[3 * i + 2]};
        if (sqrtf(dot3f(d, d)) > b.scale[i] * 2) {
            b.ok[i] = 0;
            if (b.stage) b.stage[i] = 12;
        }
    }
    if (b.n_images[i] < 0) b.n_images[i] = 0;
}

void launch_drift_gate(int n, const double* xyz, const DevBatch& b, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(drift_gate_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, xyz, b);
}

// ------------------------------------------------------------------------------------------
// Frontier expansion (reference src/hpmvs/CellProcessor.cpp:84-178 extend, :210-262 branch).
//
// Candidate directions: dx = std::cos(angle), dy = std::sin(angle) with the FLOAT angle
// (float)(2*pi/N*i) (extend, N = 6) or (float)(2*pi/N*i + pi/4) (branch, N = 4), i.e. the C library's
// cosf/sinf.  They are constants, tabulated here as the host libm (glibc 2.35) returns them;
// tests/test_cpu_abi.py checks the table against the oracle's libm values.
static const float kExtendDir[12] = {0x1p+0f, 0x0p+0f, 0x1.fffffep-2f, 0x1.bb67bp-1f, -0x1.000002p-1f, 0x1.bb67aep-1f,
                                     -0x1p+0f, -0x1.777a5cp-24f, -0x1.fffffap-2f, -0x1.bb67bp-1f, 0x1.fffffap-2f, -0x1.bb67bp-1f};
static const float kBranchDir[8] = {0x1.6a09e6p-1f, 0x1.6a09e6p-1f, -0x1.6a09e6p-1f, 0x1.6a09e6p-1f,
                                    -0x1.6a09e8p-1f, -0x1.6a09e4p-1f, 0x1.6a09e2p-1f, -0x1.6a09eap-1f};
struct DirTable { float d[12]; };

int expand_fanout(int mode) { return mode == 0 ? 6 : 4; }
const float* expand_direction_table(int mode) { return mode == 0 ? kExtendDir : kBranchDir; }

// Cell::contains (doctree.cpp:38-42)
__device__ __forceinline__ bool cell_contains(const float* cc, float width, const float* p) {
    const float hw = (float)((double)width / 2.0);
    return p[0] > cc[0] - hw && p[1] > cc[1] - hw && p[2] > cc[2] - hw && p[0] <= cc[0] + hw && p[1] <= cc[1] + hw &&
           p[2] <= cc[2] + hw;
}

// one thread per candidate (CellProcessor.cpp:93-118, 218-244)
__global__ void __launch_bounds__(256) expand_init_kernel(DevScene sc, int mode, int N, DirTable dirs, int n, DevBatch par,
                                                          const float* __restrict__ cell_center,
                                                          const float* __restrict__ cell_width,
                                                          const uint8_t* __restrict__ skip, DevBatch out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * N) return;
    const int i = t / N, k = t - i * N;
    const float pc[4] = {par.center[4 * i], par.center[4 * i + 1], par.center[4 * i + 2], par.center[4 * i + 3]};
    const float pn[4] = {par.normal[4 * i], par.normal[4 * i + 1], par.normal[4 * i + 2], par.normal[4 * i + 3]};
    const int nimg = par.n_images[i];
    const int32_t* prow = par.images + (size_t)i * par.max_images;
    int32_t* row = out.images + (size_t)t * out.max_images;
    int why = 0;
    float c[3] = {pc[0], pc[1], pc[2]};
    const float width = cell_width[i];
    if (nimg <= 0 || nimg > out.max_images || prow[0] < 0 || prow[0] >= sc.n_views) why = 20;
    else {
        const DevView& V = sc.views[prow[0]];
        float t0[3], yaxis[3], xaxis[3];
        cross3f(pn, V.xaxis, t0);
        normalized3f(t0, yaxis);
        cross3f(yaxis, pn, xaxis);
        const float dx = dirs.d[2 * k], dy = dirs.d[2 * k + 1];
        const float ext = (mode == 0) ? width : (float)((double)width / 4.0);
        for (int j = 0; j < 3; j++) c[j] = pc[j] + (dx * xaxis[j] + dy * yaxis[j]) * ext;
        for (int j = 0; j < nimg; j++) row[j] = prow[j];
        if (skip && skip[t]) why = 20;
        if (mode == 1 && !cell_contains(cell_center + 3 * i, width, c)) why = 20;
    }
    out.center[4 * t] = c[0]; out.center[4 * t + 1] = c[1]; out.center[4 * t + 2] = c[2]; out.center[4 * t + 3] = pc[3];
    out.normal[4 * t] = pn[0]; out.normal[4 * t + 1] = pn[1]; out.normal[4 * t + 2] = pn[2]; out.normal[4 * t + 3] = pn[3];
    out.scale[t] = (float)((double)width * (mode == 0 ? 0.9 : 0.45) / 2.0);
    out.n_images[t] = why ? -why : nimg;  // negative: not optimized, the refinement kernel reports -n as stage
    out.ok[t] = 0;
}

// acceptance gates after optimize() (CellProcessor.cpp:131-133, 257)
__global__ void __launch_bounds__(256) expand_gate_kernel(int mode, int N, int n, DevBatch par,
                                                          const float* __restrict__ cell_center,
                                                          const float* __restrict__ cell_width, DevBatch out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * N) return;
    const int i = t / N;
    if (out.ok[t]) {
        const float width = cell_width[i];
        const float c[3] = {out.center[4 * t], out.center[4 * t + 1], out.center[4 * t + 2]};
        int why = 0;
        if (mode == 0) {
            const double s2 = (double)out.scale[t] * 2.0;
            if (!(s2 < (double)width && s2 > (double)width / 2.0)) why = 21;
            else {
                const float d[3] = {c[0] - par.center[4 * i], c[1] - par.center[4 * i + 1], c[2] - par.center[4 * i + 2]};
                const float w = out.center[4 * t + 3] - par.center[4 * i + 3];
                const float nrm = sqrtf(dot3f(d, d) + w * w);  // Vector4f::norm()
                if (!((double)nrm < (double)width * 1.5)) why = 22;
            }
        } else if (!cell_contains(cell_center + 3 * i, width, c)) why = 22;
        if (why) {
            out.ok[t] = 0;
            if (out.stage) out.stage[t] = why;
        }
    }
    if (out.n_images[t] < 0) out.n_images[t] = 0;
}

void launch_expand_init(const DevScene& sc, int mode, int n, const DevBatch& parents, const float* cell_center,
                        const float* cell_width, const uint8_t* skip, const DevBatch& out, hipStream_t st) {
    if (n <= 0) return;
    const int N = expand_fanout(mode);
    DirTable dt;
    for (int k = 0; k < 12; k++) dt.d[k] = k < 2 * N ? expand_direction_table(mode)[k] : 0.0f;
    hipLaunchKernelGGL(expand_init_kernel, dim3((n * N + 255) / 256), dim3(256), 0, st, sc, mode, N, dt, n, parents,
                       cell_center, cell_width, skip, out);
}

void launch_expand_gate(int mode, int n, const DevBatch& parents, const float* cell_center, const float* cell_width,
                        const DevBatch& out, hipStream_t st) {
    if (n <= 0) return;
    const int N = expand_fanout(mode);
    hipLaunchKernelGGL(expand_gate_kernel, dim3((n * N + 255) / 256), dim3(256), 0, st, mode, N, n, parents, cell_center,
                       cell_width, out);
}

// ------------------------------------------------------------------------------------------
// Refined-patch records of the per-round all-gather (hpmvs_record, 192 bytes): one thread per patch, the 128 id
// bytes as eight 16-byte stores.
struct alignas(16) DevRecord {
    float center[4]; float normal[4]; float color[3]; float scale; double fmin;
    uint8_t ok, pad0; uint16_t n_images; uint8_t pad1[4];
    uint16_t images[kMaxImages];
};
static_assert(sizeof(DevRecord) == 192, "record layout");

__global__ void __launch_bounds__(256) pack_records_kernel(DevBatch b, DevRecord* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b.n) return;
    DevRecord r;
    for (int k = 0; k < 4; k++) { r.center[k] = b.center[4 * i + k]; r.normal[k] = b.normal[4 * i + k]; }
    for (int k = 0; k < 3; k++) r.color[k] = b.color ? b.color[3 * i + k] : 0.0f;
    r.scale = b.scale[i];
    r.fmin = b.fmin ? b.fmin[i] : 0.0;
    r.ok = b.ok ? b.ok[i] : 0; r.pad0 = 0;
    const int nim = b.n_images[i];
    r.n_images = (uint16_t)nim;   // (the true count; beyond kMaxImages ids only the first 64 travel: INTEGRATION.md)
    for (int k = 0; k < 4; k++) r.pad1[k] = 0;
    for (int k = 0; k < kMaxImages; k++)
        r.images[k] = (k < nim && k < b.max_images) ? (uint16_t)b.images[(size_t)i * b.max_images + k] : (uint16_t)0xFFFF;
    out[i] = r;
}

__global__ void __launch_bounds__(256) unpack_records_kernel(const DevRecord* __restrict__ in, int n, DevBatch b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DevRecord r = in[i];
    for (int k = 0; k < 4; k++) { b.center[4 * i + k] = r.center[k]; b.normal[4 * i + k] = r.normal[k]; }
    if (b.color) for (int k = 0; k < 3; k++) b.color[3 * i + k] = r.color[k];
    b.scale[i] = r.scale;
    if (b.fmin) b.fmin[i] = r.fmin;
    if (b.ok) b.ok[i] = r.ok;
    b.n_images[i] = (int32_t)(int16_t)r.n_images;
    for (int k = 0; k < b.max_images; k++)
        b.images[(size_t)i * b.max_images + k] = (k < kMaxImages && r.images[k] != 0xFFFF) ? (int32_t)r.images[k] : -1;
}

// Tails of the records whose list is longer than the record's 64 ids (include/hpmvs_amd.h: hpmvs_record_tail, 392 bytes).
// One wavefront per 64 patches: the refined patches with more than 64 images are ranked by a ballot prefix inside the wavefront
// and by an exclusive scan over the wavefronts' counts (counts[] filled by a first pass), so that the tails come out in PATCH ORDER.
struct DevTail { int32_t patch; uint16_t count, pad; uint16_t images[kWideImages - kMaxImages]; };
static_assert(sizeof(DevTail) == 392, "tail layout");
__device__ __forceinline__ bool has_tail(const DevBatch& b, int i) {
    return i < b.n && (!b.ok || b.ok[i]) && b.n_images[i] > kMaxImages && b.n_images[i] <= b.max_images && b.n_images[i] <= kWideImages;
}
__global__ void __launch_bounds__(64) tail_count_kernel(DevBatch b, int32_t* __restrict__ counts) {
    const int i = blockIdx.x * 64 + (int)threadIdx.x;
    const unsigned long long mk = __ballot(has_tail(b, i));
    if (threadIdx.x == 0) counts[blockIdx.x] = __popcll(mk);
}
__global__ void __launch_bounds__(256) tail_scan_kernel(int32_t* __restrict__ counts, int nblocks, int32_t* __restrict__ total) {
    // (one workgroup: the number of 64-patch blocks is small next to the work of a round; sequential chunks of 256)
    __shared__ int32_t part[256];
    __shared__ int32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += 256) {
        const int k = base + (int)threadIdx.x;
        const int32_t v = k < nblocks ? counts[k] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const int32_t a = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0;
            __syncthreads();
            part[threadIdx.x] += a;
            __syncthreads();
        }
        if (k < nblocks) counts[k] = carry + part[threadIdx.x] - v;   // exclusive
        __syncthreads();
        if (threadIdx.x == 255) carry += part[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ void __launch_bounds__(64) tail_pack_kernel(DevBatch b, const int32_t* __restrict__ offs, DevTail* __restrict__ out, int cap) {
    const int i = blockIdx.x * 64 + (int)threadIdx.x;
    const bool t = has_tail(b, i);
    const unsigned long long mk = __ballot(t);
    if (!t) return;
    const int pos = offs[blockIdx.x] + __popcll(mk & ((1ull << threadIdx.x) - 1ull));
    if (pos >= cap) return;
    DevTail r;
    const int nim = b.n_images[i];
    r.patch = i; r.count = (uint16_t)(nim - kMaxImages); r.pad = 0;
    for (int k = 0; k < kWideImages - kMaxImages; k++)
        r.images[k] = (kMaxImages + k < nim) ? (uint16_t)b.images[(size_t)i * b.max_images + kMaxImages + k] : (uint16_t)0xFFFF;
    out[pos] = r;
}
__global__ void __launch_bounds__(256) tail_unpack_kernel(const DevTail* __restrict__ in, int n_tails, int patch_offset, DevBatch b) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tails) return;
    const DevTail r = in[t];
    const long long i = (long long)patch_offset + r.patch;
    if (i < 0 || i >= b.n) return;
    for (int k = 0; k < (int)r.count && kMaxImages + k < b.max_images; k++)
        b.images[(size_t)i * b.max_images + kMaxImages + k] = (int32_t)r.images[k];
}
void launch_pack_record_tails(const DevBatch& b, void* tails, int cap, int32_t* scratch_counts, int32_t* total, hipStream_t st) {
    if (b.n <= 0) return;
    const int nblocks = (b.n + 63) / 64;
    hipLaunchKernelGGL(tail_count_kernel, dim3(nblocks), dim3(64), 0, st, b, scratch_counts);
    hipLaunchKernelGGL(tail_scan_kernel, dim3(1), dim3(256), 0, st, scratch_counts, nblocks, total);
    hipLaunchKernelGGL(tail_pack_kernel, dim3(nblocks), dim3(64), 0, st, b, scratch_counts, (DevTail*)tails, cap);
}
void launch_unpack_record_tails(const void* tails, int n_tails, int patch_offset, const DevBatch& b, hipStream_t st) {
    if (n_tails <= 0) return;
    hipLaunchKernelGGL(tail_unpack_kernel, dim3((n_tails + 255) / 256), dim3(256), 0, st, (const DevTail*)tails, n_tails, patch_offset, b);
}

void launch_pack_records(const DevBatch& b, void* records, hipStream_t st) {
    if (b.n <= 0) return;
    hipLaunchKernelGGL(pack_records_kernel, dim3((b.n + 255) / 256), dim3(256), 0, st, b, (DevRecord*)records);
}
void launch_unpack_records(const void* records, int n, const DevBatch& b, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(unpack_records_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const DevRecord*)records, n, b);
}

}  // namespace hpmvs
