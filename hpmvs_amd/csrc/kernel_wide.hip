// kernel_wide.hip -- PatchOptimizer::optimize for patches whose attached-image list outgrows the 64-id rows of the
// batch kernel (kernel_optimize.hip), up to HPMVS_MAX_IMAGES = 256 ids.
//
// The reference's list is an unbounded std::vector<int> (include/hpmvs/PatchOptimizer.h:106, push_back at
// src/hpmvs/PatchOptimizer.cpp:254): in a 200-view scene addImages can attach far more than 64 views before the first
// filterImagesNCC cuts the list down.  The batch kernel gives every patch slot a 64-id row (lane = list entry in its list
// services, 64-bit masks in its photometric service) and ends a patch whose list does not fit with stage 100; it marks
// such a patch with ok = 2.  THIS kernel runs behind every batch launch, scans ok[] and redoes the marked patches from
// their untouched inputs -- one wavefront per patch, the whole pipeline of runOptimization (PatchOptimizer.cpp:48-103) as
// wave-uniform sequential code:
//   * the list and everything per entry (rays, weights, flags) are LDS arrays of 256; passes run lane-strided over them with
//     the reference's float operations per entry, order-dependent parts (append order, stable compaction, greedy selection,
//     index-order sums) as ballots / ranks / uniform loops -- the wide forms of pipeline.hpp's services;
//   * a photometric request over m <= 256 images is cut into sub-requests of the reference image + up to 63 others, which the
//     PRODUCT evaluator (wave_eval_rows, photometric.hpp) serves side by side exactly as it serves the batch kernel's
//     requests; it leaves the per-image values in rows and this kernel adds them up in the reference's index order
//     (objective_fn's float64 sum, setINCCs' values, std::accumulate), so a value is the bits the 64-id path produces;
//   * the optimiser is the same bobyqa3.hpp state machine, one state in LDS, stepped by lane 0.
// Slow per patch next to the batch kernel (one patch per wavefront), exact, and only ever run for the patches that need it.
// Stage codes, diagnostics and "patch untouched on failure" as in the batch kernel.  A list that outgrows 256 ids, or a
// refined list longer than the caller's rows, still ends with stage 100.
#include <hip/hip_runtime.h>

#define BQ_STATE_LDS 1
#define BQ_P 1
#define BQ_SLOTS 1
#include "bobyqa3.hpp"
#include "launch.h"
#include "photometric.hpp"
#include "pipeline.hpp"

namespace hpmvs {

namespace {

constexpr int kWide = kWideImages;   // ids per list
constexpr int kChunk = 63;           // images per sub-request beside the reference image
constexpr int kSubs = (kWide - 1 + kChunk - 1) / kChunk;  // 5

__shared__ uint16_t w_list[kWide], w_tmp[kWide];
__shared__ uint16_t w_sub[kSubs][64];
__shared__ float w_rows[kSubs * 64];
__shared__ float w_val[kWide];      // a request's per-image values: 1 - ncc or its robust form, 2.0 for a failed grab
__shared__ uint8_t w_ok[kWide];     // ... and whether the image's grab passed its gates
__shared__ uint8_t w_flag[kWide];   // keep / candidate / alive flags of the list passes
__shared__ float w_w[kWide], w_rx[kWide], w_ry[kWide], w_rz[kWide];
__shared__ int w_pick[2];

struct WideEval { bool refok; int ngrabs; };

// sampleTexture of image `ref` of the list and of every other image, each correlated with it: w_val[k] / w_ok[k] for all k < m
// (w_val[ref] = 0).  obj: objective_fn's call (z axis = the patch's z axis, robust values); else setINCCs(ref, robust).
__device__ __noinline__ WideEval wide_eval(const DevScene& sc, const DevOptions& o, int m, int ref, int obj, int robust, const float* c,
                                           const float* zax, const float* px, const float* py, float scale) {
    const int lane = lane_id();
    const int others = m - 1;
    const int nsub = others > 0 ? (others + kChunk - 1) / kChunk : 1;
    for (int t = lane; t < others; t += 64) {
        const int k = t < ref ? t : t + 1;
        w_sub[t / kChunk][1 + t % kChunk] = w_list[k];
    }
    if (lane < nsub) w_sub[lane][0] = w_list[ref];
    wave_lds_sync();
    PackedReq q;
    // Every sub-request is posted in setINCCs' form (that is the form whose per-image values and mask of usable images come back):
    // objective_fn's correlation is the robust one against list entry 0 with the z axis it hands over -- the same grabs.
    q.obj_mode = 0; q.ref = 0; q.robust = (obj || robust) ? 1 : 0; q.thr = 0.0f; q.scale = scale;
    q.nimg = 0;
    if (lane < nsub) { const int left = others - kChunk * lane; q.nimg = 1 + (left > kChunk ? kChunk : (left > 0 ? left : 0)); }
    for (int k = 0; k < 3; k++) { q.c[k] = c[k]; q.zax[k] = zax[k]; q.px[k] = px[k]; q.py[k] = py[k]; }
    PackedRes r;
    r.obj = 2.0; r.keep = 0ull; r.incc_sum = 0.0f; r.ngrabs = 0;
    const uint16_t* mylist = &w_sub[lane < nsub ? lane : 0][0];
    wave_eval_rows<true>(sc, o, mylist, nsub >= 64 ? ~0ull : ((1ull << nsub) - 1ull), q, r, LdsRows{w_rows, 64});
    wave_lds_sync();
    unsigned long long km[kSubs];
#pragma unroll
    for (int j = 0; j < kSubs; j++) km[j] = bcast_u64(r.keep, j);   // the sub-requests' masks of usable images (bit 0: the reference image)
    WideEval e;
    e.refok = (km[0] & 1ull) != 0ull;
    e.ngrabs = 0;
    if (e.refok) {
        e.ngrabs = 1;
#pragma unroll
        for (int j = 0; j < kSubs; j++) if (j < nsub) e.ngrabs += __popcll(km[j] & ~1ull);
    }
    for (int t = lane; t < others; t += 64) {
        const int k = t < ref ? t : t + 1;
        const int j = t / kChunk, i = 1 + t % kChunk;
        unsigned long long mk = km[0];
#pragma unroll
        for (int jj = 1; jj < kSubs; jj++) mk = (j == jj) ? km[jj] : mk;
        const bool okk = e.refok && ((mk >> i) & 1ull);
        w_ok[k] = okk ? 1 : 0;
        w_val[k] = okk ? w_rows[j * 64 + i] : 2.0f;
    }
    if (lane == 0) { w_ok[ref] = e.refok ? 1 : 0; w_val[ref] = e.refok ? 0.0f : 2.0f; }
    wave_lds_sync();
    return e;
}

// stable compaction of w_list by w_flag; returns the new count
__device__ __forceinline__ int wide_compact(int m) {
    const int lane = lane_id();
    int base = 0;
    for (int j0 = 0; j0 < m; j0 += 64) {
        const int k = j0 + lane;
        const bool kp = k < m && w_flag[k] != 0;
        const unsigned long long mask = __ballot(kp);
        if (kp) w_tmp[base + __popcll(mask & lanes_below(lane))] = w_list[k];
        base += __popcll(mask);
    }
    wave_lds_sync();
    for (int k = lane; k < base; k += 64) w_list[k] = w_tmp[k];
    wave_lds_sync();
    return base;
}

// PatchOptimizer::addImages (PatchOptimizer.cpp:225-258) on w_list; the new count, -1 when the list would outgrow kWide
__device__ __noinline__ int wide_add_images(ListEnv le, int m, const float* c, const float* n, float scale) {
    if (m <= 0) return 0;
    const int lane = lane_id();
    const int ref = (int)w_list[0];
    const int existing = m;
    const int beg = le.covis_off[ref], end = le.covis_off[ref + 1];
    for (int base = beg; base < end; base += 64) {
        const int ci = base + lane;
        const bool valid = ci < end;
        const int cv = valid ? le.covis[ci] : -2;
        bool found = false;
        for (int k = 0; k < existing; k++) found |= ((int)w_list[k] == cv);
        bool pass = false;
        if (valid && !found) pass = add_images_test(le.views[cv], le, c, n, scale);
        const unsigned long long mk = __ballot(pass);
        const int cnt = __popcll(mk);
        if (m + cnt > kWide) return -1;
        if (pass) w_list[m + __popcll(mk & lanes_below(lane))] = (uint16_t)cv;
        m += cnt;
        wave_lds_sync();
    }
    return m;
}

// the rays (camera centre - patch centre).normalized() of the list's images into w_rx / w_ry / w_rz
__device__ __forceinline__ void wide_rays(const ListEnv& le, int m, const float* c) {
    for (int k = lane_id(); k < m; k += 64) {
        float ray[3];
        ray_to_cam(le.views[w_list[k]], c, ray);
        w_rx[k] = ray[0]; w_ry[k] = ray[1]; w_rz[k] = ray[2];
    }
    wave_lds_sync();
}

// PatchOptimizer::sortImages + getAngleWeightedScales (PatchOptimizer.cpp:183-223, 260-284); the new count
__device__ __noinline__ int wide_sort_images(ListEnv le, int m, const float* c, const float* n, float scale) {
    const int lane = lane_id();
    if (m <= 0) return 0;
    const DevView& V0 = le.views[w_list[0]];
    const float dd[3] = {c[0] - V0.center[0], c[1] - V0.center[1], c[2] - V0.center[2]};
    const float fz0 = sqrtf(dot3f(dd, dd));
    int refLevel = cvt_i32_x86(roundf((float)log2((double)(scale * V0.fsum) / (2.0 * (double)fz0))));
    refLevel = refLevel < le.MAXLEVEL - 1 ? refLevel : le.MAXLEVEL - 1;
    refLevel = refLevel > 0 ? refLevel : 0;
    float nn[3];
    normalized3f(n, nn);
    int alive = 0;
    for (int j0 = 0; j0 < m; j0 += 64) {
        const int k = j0 + lane;
        bool cand = false;
        if (k < m) {
            const DevView& V = le.views[w_list[k]];
            float ray[3];
            ray_to_cam(V, c, ray);
            const float cosa = dot3f(ray, nn);
            float w = 0.0f;
            if (cosa > 0.0f) {
                cand = true;
                const float e[3] = {c[0] - V.center[0], c[1] - V.center[1], c[2] - V.center[2]};
                const float fz = sqrtf(dot3f(e, e));
                const float sc_ = (V.fsum == 0.0f) ? 1.0f : (float)(2.0 * (double)fz * (double)(1 << refLevel) / (double)V.fsum);
                w = sc_ / cosa;
            }
            w_w[k] = w; w_rx[k] = ray[0]; w_ry[k] = ray[1]; w_rz[k] = ray[2];
            w_flag[k] = cand ? 1 : 0;
        }
        alive += __popcll(__ballot(cand));
    }
    wave_lds_sync();
    if (alive < 2) return 0;  // pImages_ was cleared
    {   // the first candidate keeps its place
        int first = -1;
        for (int j0 = 0; j0 < m && first < 0; j0 += 64) {
            const int k = j0 + lane;
            const unsigned long long mk = __ballot(k < m && w_flag[k] != 0);
            if (mk) first = j0 + __ffsll((long long)mk) - 1;
        }
        if (lane == 0) w_w[first] = 0.0f;
        wave_lds_sync();
    }
    const float thr = le.sort_thr;
    int out = 0;
    while (alive > 0) {
        // greedy pick: the smallest weight, the earliest entry among equals
        float mine = __int_as_float(0x7f800000);
        for (int k = lane; k < m; k += 64) if (w_flag[k]) mine = fminf(mine, w_w[k]);
        const float mn = wave_min_f32(mine);
        int best = -1, firstalive = -1;
        for (int j0 = 0; j0 < m; j0 += 64) {
            const int k = j0 + lane;
            const bool al = k < m && w_flag[k] != 0;
            const unsigned long long eq = __ballot(al && w_w[k] == mn), am = __ballot(al);
            if (best < 0 && eq) best = j0 + __ffsll((long long)eq) - 1;
            if (firstalive < 0 && am) firstalive = j0 + __ffsll((long long)am) - 1;
        }
        if (best < 0) best = firstalive;  // (weights that are all NaN compare equal to nothing: the earliest remaining entry)
        const float rb[3] = {w_rx[best], w_ry[best], w_rz[best]};
        wave_lds_sync();
        if (lane == 0) { w_tmp[out] = w_list[best]; w_flag[best] = 0; }
        out++;
        alive--;
        wave_lds_sync();
        for (int k = lane; k < m; k += 64)
            if (w_flag[k]) {
                const float ray[3] = {w_rx[k], w_ry[k], w_rz[k]};
                const float ftmp = fminf(thr, fmaxf(thr / 2.0f, 1.0f - dot3f(rb, ray)));
                w_w[k] = w_w[k] * (thr / ftmp);
            }
        wave_lds_sync();
    }
    for (int k = lane; k < out; k += 64) w_list[k] = w_tmp[k];
    wave_lds_sync();
    return out;
}

// PatchOptimizer::assureImageAngles (PatchOptimizer.cpp:105-123)
__device__ __noinline__ bool wide_assure_image_angles(ListEnv le, int m, const float* c) {
    const int lane = lane_id();
    wide_rays(le, m, c);
    for (int ii = 0; ii < m - 1; ii++) {
        const float ri[3] = {w_rx[ii], w_ry[ii], w_rz[ii]};
        bool hit = false;
        for (int k = lane; k < m; k += 64)
            if (k > ii) {
                const float rj[3] = {w_rx[k], w_ry[k], w_rz[k]};
                const float a = acosf(dot3f(ri, rj));
                hit |= a < le.MAX_ANGLE && a > le.MIN_ANGLE;
            }
        if (__any(hit)) return true;
    }
    return false;
}

// PatchOptimizer::filterImagesByAngle (PatchOptimizer.cpp:125-136)
__device__ __forceinline__ int wide_filter_by_angle(const ListEnv& le, int m, const float* c, const float* n) {
    wide_rays(le, m, c);
    for (int k = lane_id(); k < m; k += 64) {
        const float ray[3] = {w_rx[k], w_ry[k], w_rz[k]};
        w_flag[k] = dot3f(ray, n) > le.cos_max_f ? 1 : 0;
    }
    wave_lds_sync();
    return wide_compact(m);
}

// Scene::getColor(const Patch3d&) (Scene.cpp:300-327): the median colour by norm, the darkest if that one is brighter than 250
__device__ __noinline__ U3 wide_patch_colour(const DevScene& sc, int m, const float* c, float scale) {
    const int lane = lane_id();
    for (int k = lane; k < m; k += 64) {
        float rgb[3];
        colour_at(sc, sc.views[w_list[k]], c, scale, rgb);
        w_rx[k] = rgb[0]; w_ry[k] = rgb[1]; w_rz[k] = rgb[2];
        w_w[k] = sqrtf(dot3f(rgb, rgb));
    }
    if (lane == 0) { w_pick[0] = 0; w_pick[1] = 0; }
    wave_lds_sync();
    const int want = m / 2;
    for (int k = lane; k < m; k += 64) {
        const float nk = w_w[k];
        int rank = 0;
        for (int j = 0; j < m; j++) { const float nj = w_w[j]; rank += (nj < nk) || (nj == nk && j < k); }
        if (rank == want) w_pick[0] = k;
        if (rank == 0) w_pick[1] = k;
    }
    wave_lds_sync();
    const int med = w_pick[0], mn = w_pick[1];
    const int pick = (w_w[med] > 250.0f) ? mn : med;
    return U3{w_rx[pick], w_ry[pick], w_rz[pick]};
}

// filterImagesNCC (PatchOptimizer.cpp:138-152): setINCCs(ref = 0, robust = 0), keep image 0 and those with 1 - ncc < 1 - thr
__device__ __forceinline__ int wide_filter_ncc(const DevScene& sc, const DevOptions& o, int m, float thr, const float* c, const float* n,
                                               float scale, int& ngrabs) {
    float ax[3], ay[3], az[3];
    const DevView& RV = sc.views[w_list[0]];
    calc_patch_axes(RV.xaxis, RV.yaxis, n, scale, ax, ay, az);
    const WideEval e = wide_eval(sc, o, m, 0, 0, 0, c, n, ax, ay, scale);   // (setINCCs hands pNormal_, not the z axis, to sampleTexture)
    ngrabs += e.ngrabs;
    const float thr1 = 1.0f - thr;
    for (int k = lane_id(); k < m; k += 64) w_flag[k] = (k == 0) ? 1 : ((e.refok && w_val[k] < thr1) ? 1 : 0);
    wave_lds_sync();
    return wide_compact(m);
}

struct WideOut { int ok, stage, nimg, nl_result, ngrabs, nevals; float c[3], n[3], rgb[3]; double fmin, x[3]; };

// runOptimization (PatchOptimizer.cpp:48-76) + optimize()'s copy-out (:78-103) for the patch whose list is in w_list
__device__ void wide_run(const DevScene& sc, const DevOptions& o, int m, const float* c0, const float* n0, float scale, int max_images,
                         bool want_colour, WideOut& R) {
    const int lane = lane_id();
    const ListEnv le = list_env(sc, o);
    float c[3] = {c0[0], c0[1], c0[2]}, n[3] = {n0[0], n0[1], n0[2]};
    R.ok = 0; R.stage = 0; R.nimg = 0; R.nl_result = 0; R.ngrabs = 0; R.nevals = 0; R.fmin = 0.0;
    for (int k = 0; k < 3; k++) { R.x[k] = 0.0; R.rgb[k] = 0.0f; }
    const int MIN = o.MIN_IMAGES_PER_PATCH;
#define WIDE_FAIL(code) do { R.stage = (code); return; } while (0)
    m = wide_add_images(le, m, c, n, scale);
    if (m < 0) WIDE_FAIL(100);
    if (m < MIN) WIDE_FAIL(1);
    m = wide_filter_ncc(sc, o, m, o.NCC_ALPHA_1, c, n, scale, R.ngrabs);
    if (m < MIN) WIDE_FAIL(2);
    m = wide_sort_images(le, m, c, n, scale);
    if (!wide_assure_image_angles(le, m, c)) WIDE_FAIL(3);
    if (m < MIN) WIDE_FAIL(4);
    {   // optimizePatch (PatchOptimizer.cpp:322-382)
        OptFields F;
        set_optimization_fields(sc.views[w_list[0]], c, F);
        const double lbv[3] = {-HUGE_VAL, -23.99999, -23.99999}, ubv[3] = {HUGE_VAL, 23.99999, 23.99999};
        double x0[3] = {0, 0, 0}, dx[3] = {1, 1, 1};
        parameters_from_center_norm(F, o, n, lbv, ubv, x0);
        bq_default_step(x0, lbv, ubv, dx);
        Bobyqa3 B;
        B.slot = 0; B.slot_i = 0;
        int ret = BQ_DONE;
        if (lane == 0) ret = bq_start(B, x0, lbv, ubv, dx, 1.e-7, 1000);
        ret = bcast_i(ret, 0);
        if (ret != BQ_CONTINUE) WIDE_FAIL(4);   // bq_start refused the arguments
        const DevView& RV = sc.views[w_list[0]];
        for (;;) {
            wave_lds_sync();
            const double xe[3] = {BQA(xeval, 0), BQA(xeval, 1), BQA(xeval, 2)};   // (one state: every lane reads the same words)
            set_center_norm(F, o, xe, c, n);
            if (ret != BQ_CONTINUE) break;
            // objective_fn (PatchOptimizer.cpp:286-311)
            float ax[3], ay[3], az[3];
            calc_patch_axes(RV.xaxis, RV.yaxis, n, scale, ax, ay, az);
            const WideEval e = wide_eval(sc, o, m, 0, 1, 0, c, az, ax, ay, scale);
            R.ngrabs += e.ngrabs;
            double f = 2.0;
            if (e.refok) {
                double val = 0.0;
                int nok = 0;
                for (int k = 1; k < m; k++)
                    if (w_ok[k]) { val += (double)w_val[k]; nok++; }
                f = (nok < MIN - 1) ? 2.0 : val / (double)nok;
            }
            if (lane == 0) ret = bq_step(B, f);
            ret = bcast_i(ret, 0);
        }
        wave_lds_sync();
        R.nl_result = BQN(rc); R.fmin = BQS(minf); R.nevals = BQN(nevals);
        for (int k = 0; k < 3; k++) R.x[k] = BQA(xeval, k);
        if (!(R.nl_result >= 1 && R.nl_result <= 4)) WIDE_FAIL(4);   // PatchOptimizer.cpp:367-368; negative codes throw
    }
    m = wide_add_images(le, m, c, n, scale);
    if (m < 0) WIDE_FAIL(100);
    if (m < MIN) WIDE_FAIL(5);
    m = wide_filter_ncc(sc, o, m, o.NCC_ALPHA_2, c, n, scale, R.ngrabs);
    if (m < MIN) WIDE_FAIL(6);
    m = wide_filter_by_angle(le, m, c, n);
    if (m < MIN) WIDE_FAIL(7);
    if (!wide_assure_image_angles(le, m, c)) WIDE_FAIL(8);
    {   // setRefImage (PatchOptimizer.cpp:154-181): one robust setINCCs per candidate reference image
        int refindex = -1;
        float refbest = 3.402823466e+38f;
        for (int ii = 0; ii < m; ii++) {
            float ax[3], ay[3], az[3];
            const DevView& RV = sc.views[w_list[ii]];
            calc_patch_axes(RV.xaxis, RV.yaxis, n, scale, ax, ay, az);
            const WideEval e = wide_eval(sc, o, m, ii, 0, 1, c, n, ax, ay, scale);
            R.ngrabs += e.ngrabs;
            float s = 0.0f;
            if (e.refok) { for (int k = 0; k < m; k++) if (k != ii) s += w_val[k]; }
            else for (int k = 0; k < m; k++) s += 2.0f;
            if (s < refbest) { refbest = s; refindex = ii; }
        }
        if (refindex > 0 && lane == 0) { const uint16_t t = w_list[0]; w_list[0] = w_list[refindex]; w_list[refindex] = t; }
        wave_lds_sync();
    }
    m = wide_filter_ncc(sc, o, m, o.NCC_ALPHA_2, c, n, scale, R.ngrabs);
    if (m < MIN) WIDE_FAIL(9);
    if (m > max_images) WIDE_FAIL(100);   // the refined list does not fit the caller's rows
#undef WIDE_FAIL
    if (want_colour) { const U3 col = wide_patch_colour(sc, m, c, scale); R.rgb[0] = col.x; R.rgb[1] = col.y; R.rgb[2] = col.z; }
    R.ok = 1; R.nimg = m;
    for (int k = 0; k < 3; k++) { R.c[k] = c[k]; R.n[k] = n[k]; }
}

}  // namespace

// Two launches behind the batch kernel.  wide_collect_kernel: one thread per patch, the marked ones (ok == 2) are appended to a
// list (one atomic per wavefront that has any); the list lives in the launch workspace behind the counter block -- the batch
// kernel's parking buffers, which nobody needs any more when this runs.  optimize_wide_kernel: one wavefront per workgroup,
// each takes the next list entry until the list is through (a marked patch costs milliseconds: one patch per claim).
// counter[0]: next list entry, counter[1]: patches redone (hpmvs_last_wide_patches), counter[2]: list length.
__global__ void __launch_bounds__(256) wide_collect_kernel(DevBatch b, int32_t* counter, int32_t* list) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool marked = i < b.n && b.ok[i] == 2;
    const unsigned long long mk = __ballot(marked);
    if (!mk) return;
    const int lane = lane_id();
    int base = 0;
    if (lane == __ffsll((long long)mk) - 1) base = atomicAdd(counter + 2, __popcll(mk));
    base = bcast_i(base, __ffsll((long long)mk) - 1);
    if (marked) list[base + __popcll(mk & lanes_below(lane))] = i;
}

__global__ void __launch_bounds__(256) wide_unmark_kernel(DevBatch b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < b.n && b.ok[i] == 2) b.ok[i] = 0;
}

__global__ void __launch_bounds__(64, 2) optimize_wide_kernel(DevScene sc, DevOptions o, DevBatch b, int32_t* counter, const int32_t* list) {
    const int lane = lane_id();
    const int total = counter[2];
    for (;;) {
        int t = 0;
        if (lane == 0) t = atomicAdd(counter, 1);
        t = bcast_i(t, 0);
        if (t >= total) return;
        {
            {
                const int idx = list[t];
                const float c0[3] = {b.center[4 * idx], b.center[4 * idx + 1], b.center[4 * idx + 2]};
                const float n0[3] = {b.normal[4 * idx], b.normal[4 * idx + 1], b.normal[4 * idx + 2]};
                const float scale = b.scale[idx];
                const int nimg = b.n_images[idx];
                WideOut R;
                R.ok = 0; R.stage = 100; R.nimg = 0; R.nl_result = 0; R.ngrabs = 0; R.nevals = 0; R.fmin = 0.0;
                for (int k = 0; k < 3; k++) { R.x[k] = 0.0; R.rgb[k] = 0.0f; R.c[k] = c0[k]; R.n[k] = n0[k]; }
                if (nimg >= 0 && nimg <= kWide && nimg <= b.max_images) {
                    bool bad = false;
                    for (int k = lane; k < nimg; k += 64) {
                        const int id = b.images[(size_t)idx * b.max_images + k];
                        bad |= (id < 0 || id >= sc.n_views);
                        w_list[k] = (uint16_t)id;
                    }
                    wave_lds_sync();
                    if (__any(bad)) R.stage = 101;   // image id outside the scene: rejected, never dereferenced
                    else wide_run(sc, o, nimg, c0, n0, scale, b.max_images, b.color != nullptr, R);
                }
                wave_lds_sync();
                if (lane == 0) {
                    b.ok[idx] = R.ok ? 1 : 0;
                    if (R.ok) {
                        for (int k = 0; k < 3; k++) { b.center[4 * idx + k] = R.c[k]; b.normal[4 * idx + k] = R.n[k]; }
                        b.center[4 * idx + 3] = 1.0f; b.normal[4 * idx + 3] = 0.0f;
                        b.n_images[idx] = R.nimg;
                        if (b.color) { b.color[3 * idx] = R.rgb[0]; b.color[3 * idx + 1] = R.rgb[1]; b.color[3 * idx + 2] = R.rgb[2]; }
                        if (b.ncc) b.ncc[idx] = 1.4f;  // PatchOptimizer.cpp:95
                    }
                    if (b.stage) b.stage[idx] = R.ok ? 0 : R.stage;
                    if (b.result) b.result[idx] = R.nl_result;
                    if (b.ngrabs) b.ngrabs[idx] = R.ngrabs;
                    if (b.fmin) b.fmin[idx] = R.fmin;
                    if (b.x) for (int k = 0; k < 3; k++) b.x[3 * idx + k] = R.x[k];
                    if (b.nevals) b.nevals[idx] = R.nevals;
                }
                if (lane == 0) atomicAdd(counter + 1, 1);   // (patches this kernel redid: hpmvs_last_wide_patches)
                if (R.ok)
                    for (int k = lane; k < R.nimg; k += 64) b.images[(size_t)idx * b.max_images + k] = (int32_t)w_list[k];
                wave_lds_sync();
            }
        }
    }
}

// workspace: the launch workspace of the batch kernel that ran before (its first kCounterBytes = 1 KB are the counter block; the
// words from `counter` on are zero and nobody else's), list_cap: ints that fit behind the counter block
void launch_optimize_wide(const DevScene& sc, const DevOptions& o, const DevBatch& b, int32_t* workspace, int32_t* counter, size_t list_cap,
                          int n_cus, hipStream_t st) {
    if (b.n <= 0) return;
    if ((size_t)b.n > list_cap) {   // (more patches than the workspace holds list entries -- ~50 M: no wide pass, the marked patches keep stage 100)
        hipLaunchKernelGGL(wide_unmark_kernel, dim3((b.n + 255) / 256), dim3(256), 0, st, b);
        return;
    }
    int32_t* list = workspace + 256;
    hipLaunchKernelGGL(wide_collect_kernel, dim3((b.n + 255) / 256), dim3(256), 0, st, b, counter, list);
    int wgs = b.n < n_cus * 8 ? b.n : n_cus * 8;
    hipLaunchKernelGGL(optimize_wide_kernel, dim3(wgs), dim3(64), 0, st, sc, o, b, counter, list);
}

}  // namespace hpmvs
