// kernel_optimize.hip -- the batched PatchOptimizer::optimize kernel for gfx950.
//
// Mapping (MI355X-first, not a translation of the reference's one-thread-per-patch OpenMP loop):
//
//   * A persistent grid of single-wavefront workgroups (8 per CU, bounded by LDS) pulls patches from a
//     global work queue.
//   * Lanes 0..kP-1 of a wave each own one patch ("slot") and run that patch's control logic as
//     plain SIMT code: the optimize() pipeline of the reference
//     (src/hpmvs/PatchOptimizer.cpp:48-103: addImages / filterImagesNCC / sortImages /
//     assureImageAngles / optimizePatch / ... / setRefImage) and the BOBYQA state machine
//     (bobyqa3.hpp).  This is "phase A"; it ends when the lane needs photometric data.
//   * Photometric work is done by the WHOLE WAVE for all posted requests at once ("phase B",
//     wave_eval_rows in photometric.hpp): the (request, image) pairs become jobs, one lane each
//     prepares its image; then nine 7-lane groups (one lane per window ROW) each serve one request,
//     grab by grab, forming the reference's sequential float32 sums as DPP chains through the
//     group's lanes.  No LDS, no barriers; the results go back to the requesting lanes.
//
//   Patches finish at different times (40-1000 evaluations); a finished lane immediately pulls the
//   next patch from the queue, so slots stay busy until the queue drains.
//
//   The per-patch optimiser state (177 float64 + 13 int32) and image lists (uint16 ids, padded
//   rows) live in LDS as [index][slot], so the kP lanes touch consecutive words and none of it goes
//   through scratch / HBM.  (A first version kept the state in per-lane scratch for 64 patches per
//   wave: PMC showed 2.1 G scratch/flat instructions and 157 GB of HBM traffic per 1e5-patch launch
//   against 13 GB of algorithmic image bytes -- profiles/r01a_scratch_state.)
#include <hip/hip_runtime.h>

// optimiser state of the kP patches a wavefront owns lives in LDS (see bobyqa3.hpp)
#include "prof.h"
#define BQ_STATE_LDS 1
#ifndef BQ_P
#define BQ_P 15
#endif
// Single-wavefront workgroups: 15 x 1364 B of optimiser state = 20.0 KB of LDS (exactly 1/8 of a CU), 8 per CU; the photometric
// evaluator (wave_eval_rows) needs no LDS.  (The image lists, touched a dozen times per patch, live in a
// global-memory workspace.  Multi-wavefront workgroups -- sharing staging buffers under a lock, or scheduling the
// BOBYQA blocks of all their slots cooperatively -- were measured 5-25 % slower in round 1)
#define BQ_SLOTS BQ_P
#include "bobyqa3.hpp"
#include "launch.h"
#include "photometric.hpp"
#include "pipeline.hpp"

namespace hpmvs {

namespace {

constexpr int kP = BQ_P;  // patches ("slots") per wavefront: lanes 0..kP-1 own one each
constexpr int kRow = 64;  // uint16 ids per image-list row (global workspace, one row per slot)
constexpr size_t kCounterBytes = 1024;  // head of the launch workspace: work-queue counter (+ profiling counters)

}  // namespace

#ifndef HPMVS_WAVES_PER_EU
#define HPMVS_WAVES_PER_EU 2
#endif
__global__ void __launch_bounds__(64, HPMVS_WAVES_PER_EU) optimize_kernel(DevScene sc, DevOptions o, DevBatch b, int32_t* queue) {
    const int lane = lane_id();
    // this wavefront's image-list rows in the launch workspace (behind the counter block)
    uint16_t* lists = (uint16_t*)((char*)queue + kCounterBytes) + (size_t)blockIdx.x * kP * kRow;
    uint16_t* row = &lists[(lane < kP ? lane : 0) * kRow];

    // ---- per-lane slot state
    // Small batches are spread over as many wavefronts as possible (a wavefront with fewer busy slots has
    // shorter rounds: one lone patch advances ~2.4x faster than one of ten), so only `quota` lanes pull work.
    const int n_waves = (int)gridDim.x;
    const int quota = (b.n + n_waves - 1) / n_waves < kP ? (b.n + n_waves - 1) / n_waves : kP;
    int stage = (lane < quota) ? S_FETCH : S_IDLE;  // the other lanes only help in phase B
    int idx = -1, nimg = 0, ii = 0, refindex = -1, ngrabs = 0, fail_stage = 0, nl_result = 0;
    float c0[3] = {0, 0, 0}, n0[3] = {0, 0, 0};  // optimize() input (kept for "patch untouched")
    float c[3] = {0, 0, 0}, n[3] = {0, 0, 1};    // pCenter_, pNormal_
    float scale = 1.0f, refbest = 0.0f;
    OptFields F;
    Bobyqa3 B;
    B.slot = lane < kP ? lane : 0;
    // request posted to phase B (with the patch axes of calculatePatchAxis for the request's reference image)
    float ax[3] = {0, 0, 0}, ay[3] = {0, 0, 0}, az[3] = {0, 0, 1};
    int op = OP_NONE, req_ref = 0, req_robust = 0;
    float req_thr = 0.0f;
    // results delivered by phase B
    double res_f = 0.0;
    unsigned long long res_mask = 0ull;
    float res_sum = 0.0f;
    const double lbv[3] = {-HUGE_VAL, -23.99999, -23.99999};  // PatchOptimizer.cpp:326-336
    const double ubv[3] = {HUGE_VAL, 23.99999, 23.99999};

#ifdef HPMVS_PROFILE_PHASES
    long long tA = 0, tB = 0, nrounds = 0, nreq = 0, t0 = clock64();
    s_prof[lane] = 0ull;
    __syncthreads();
#endif
    // results delivered by the list services (pipeline.hpp)
    int res_n = 0, res_fail = 0;
    float res_rgb[3] = {0, 0, 0};
    for (;;) {
        // ================= phase A: every lane advances its own patch until it needs the wave
        while (op == OP_NONE && stage != S_IDLE) {
            bool finish_fail = false, finish_ok = false;
            switch (stage) {
            case S_FETCH: {
                idx = atomicAdd(queue, 1);
                if (idx >= b.n) { stage = S_IDLE; break; }
                for (int k = 0; k < 3; k++) { c0[k] = b.center[4 * idx + k]; n0[k] = b.normal[4 * idx + k]; c[k] = c0[k]; n[k] = n0[k]; }
                scale = b.scale[idx];
                nimg = b.n_images[idx];
                ngrabs = 0; nl_result = 0; BQN(nevals) = 0; BQS(minf) = 0.0;
                BQA(xfinal, 0) = 0.0; BQA(xfinal, 1) = 0.0; BQA(xfinal, 2) = 0.0;
                if (nimg < 0) { fail_stage = -nimg; nimg = 0; finish_fail = true; break; }  // rejected by seed_init_kernel
                if (nimg > kMaxImages || nimg > b.max_images) { fail_stage = 100; finish_fail = true; break; }
                bool bad_id = false;
                for (int k = 0; k < nimg; k++) {
                    const int id = b.images[(size_t)idx * b.max_images + k];
                    bad_id |= (id < 0 || id >= sc.n_views);
                    row[k] = (uint16_t)id;
                }
                if (bad_id) { fail_stage = 101; finish_fail = true; break; }  // image id outside the scene: rejected, never dereferenced
                op = OPL_ADD;  // addImages
                stage = S_ADD1;
                break;
            }
            case S_ADD1: {
                nimg = res_n;
                if (nimg < 0) { fail_stage = 100; finish_fail = true; break; }
                if (nimg < o.MIN_IMAGES_PER_PATCH) { fail_stage = 1; finish_fail = true; break; }
                op = OP_INCC; req_ref = 0; req_robust = 0; req_thr = o.NCC_ALPHA_1;
                stage = S_F1;
                break;
            }
            case S_F1:  // filterImagesNCC(alpha1) result -> the removal, sortImages, assureImageAngles
                op = OPL_F1;
                stage = S_F1B;
                break;
            case S_F1B: {  // optimizePatch start
                nimg = res_n;
                if (res_fail) { fail_stage = res_fail; finish_fail = true; break; }
                set_optimization_fields(sc.views[row[0]], c, F);
                double x0[3], dx[3];
                parameters_from_center_norm(F, o, n, lbv, ubv, x0);
                bq_default_step(x0, lbv, ubv, dx);
                if (bq_start(B, x0, lbv, ubv, dx, 1.e-7, 1000) != BQ_CONTINUE) {
                    nl_result = BQN(rc); fail_stage = 4; finish_fail = true; break;
                }
                { const double xe[3] = {BQA(xeval, 0), BQA(xeval, 1), BQA(xeval, 2)}; set_center_norm(F, o, xe, c, n); }
                op = OP_OBJ;
                stage = S_BQ;
                break;
            }
            case S_BQ: {  // objective value arrived: advance BOBYQA
                if (bq_step(B, res_f) == BQ_CONTINUE) {
                    const double xe[3] = {BQA(xeval, 0), BQA(xeval, 1), BQA(xeval, 2)};
                    set_center_norm(F, o, xe, c, n);
                    op = OP_OBJ;
                    break;
                }
                nl_result = BQN(rc);
                // SUCCESS / STOPVAL / FTOL / XTOL only (PatchOptimizer.cpp:367-368); negative codes throw
                if (!(nl_result >= 1 && nl_result <= 4)) { fail_stage = 4; finish_fail = true; break; }
                { const double xf[3] = {BQA(xfinal, 0), BQA(xfinal, 1), BQA(xfinal, 2)}; set_center_norm(F, o, xf, c, n); }
                op = OPL_ADD;  // addImages
                stage = S_ADD2;
                break;
            }
            case S_ADD2: {
                nimg = res_n;
                if (nimg < 0) { fail_stage = 100; finish_fail = true; break; }
                if (nimg < o.MIN_IMAGES_PER_PATCH) { fail_stage = 5; finish_fail = true; break; }
                op = OP_INCC; req_ref = 0; req_robust = 0; req_thr = o.NCC_ALPHA_2;
                stage = S_F2;
                break;
            }
            case S_F2:  // filterImagesNCC(alpha2) result -> the removal, filterImagesByAngle, assureImageAngles
                op = OPL_F2;
                stage = S_F2B;
                break;
            case S_F2B: {  // setRefImage start
                nimg = res_n;
                if (res_fail) { fail_stage = res_fail; finish_fail = true; break; }
                ii = 0; refindex = -1; refbest = 3.402823466e+38f;
                op = OP_REFSUM; req_ref = 0; req_robust = 1;
                stage = S_RS;
                break;
            }
            case S_RS: {  // setRefImage (PatchOptimizer.cpp:154-181): one robust setINCCs per candidate
                if (res_sum < refbest) { refbest = res_sum; refindex = ii; }
                ii++;
                if (ii < nimg) { op = OP_REFSUM; req_ref = ii; req_robust = 1; break; }
                if (refindex > 0) {  // swap the chosen image into slot 0
                    const uint16_t t = row[0];
                    row[0] = row[refindex];
                    row[refindex] = t;
                }
                op = OP_INCC; req_ref = 0; req_robust = 0; req_thr = o.NCC_ALPHA_2;
                stage = S_F3;
                break;
            }
            case S_F3:  // the last filterImagesNCC(alpha2) result -> the removal, Scene::getColor(patch)
                op = OPL_F3;
                stage = S_F3B;
                break;
            case S_F3B: {
                nimg = res_n;
                if (res_fail) { fail_stage = res_fail; finish_fail = true; break; }
                finish_ok = true;
                break;
            }
            default: stage = S_IDLE; break;
            }
            if (finish_ok || finish_fail) {
                b.ok[idx] = finish_ok ? 1 : 0;
                if (finish_ok) {
                    for (int k = 0; k < 3; k++) { b.center[4 * idx + k] = c[k]; b.normal[4 * idx + k] = n[k]; }
                    b.center[4 * idx + 3] = 1.0f;
                    b.normal[4 * idx + 3] = 0.0f;
                    b.n_images[idx] = nimg;
                    for (int k = 0; k < nimg && k < b.max_images; k++) b.images[(size_t)idx * b.max_images + k] = row[k];
                    if (b.color) { b.color[3 * idx] = res_rgb[0]; b.color[3 * idx + 1] = res_rgb[1]; b.color[3 * idx + 2] = res_rgb[2]; }
                    if (b.ncc) b.ncc[idx] = 1.4f;  // PatchOptimizer.cpp:95
                }
                if (b.stage) b.stage[idx] = finish_ok ? 0 : fail_stage;
                if (b.fmin) b.fmin[idx] = BQS(minf);
                if (b.x) for (int k = 0; k < 3; k++) b.x[3 * idx + k] = BQA(xfinal, k);
                if (b.result) b.result[idx] = nl_result;
                if (b.nevals) b.nevals[idx] = BQN(nevals);
                if (b.ngrabs) b.ngrabs[idx] = ngrabs;
                stage = S_FETCH;
            }
        }
        // ================= list services: the image-list passes, one request at a time, 64 lanes wide (pipeline.hpp)
        {
            wave_lds_sync();  // the lanes' own list writes (global workspace) before the cooperative reads
            const unsigned long long lm = __ballot(op >= OPL_ADD);
            if (lm) {
                BQ_PROF_BEGIN(tls_);
                for (unsigned long long t = lm; t; t &= t - 1ull) {
                    const int p = __ffsll((long long)t) - 1;
                    const int kind = bcast_i(op, p);
                    uint16_t* prow = lists + p * kRow;
                    const int pn = bcast_i(nimg, p);
                    const U3 pc{bcast_f(c[0], p), bcast_f(c[1], p), bcast_f(c[2], p)};
                    const U3 pnrm{bcast_f(n[0], p), bcast_f(n[1], p), bcast_f(n[2], p)};
                    const float pscale = bcast_f(scale, p);
                    const unsigned long long pmask = bcast_u64(res_mask, p);
                    ListResult r;
                    r.nimg = 0; r.fail = 0; r.rgb = U3{0.0f, 0.0f, 0.0f};
                    if (kind == OPL_ADD) r.nimg = coop_add_images(sc, o, prow, pn, pc, pnrm, pscale);
                    else if (kind == OPL_F1) r = coop_list_f1(sc, o, prow, pn, pmask, pc, pnrm, pscale);
                    else if (kind == OPL_F2) r = coop_list_f2(sc, o, prow, pn, pmask, pc, pnrm);
                    else r = coop_list_f3(sc, o, prow, pn, pmask, pc, pscale, b.max_images, b.color != nullptr);
                    if (lane == p) {
                        res_n = r.nimg; res_fail = r.fail;
                        res_rgb[0] = r.rgb.x; res_rgb[1] = r.rgb.y; res_rgb[2] = r.rgb.z;
                        op = OP_NONE;
                    }
                }
                wave_lds_sync();
                BQ_PROF_END(tls_, 24);
                continue;  // the served lanes go on in phase A
            }
        }
        BQ_PROF_BEGIN(tax_);
        if (op != OP_NONE) {  // calculatePatchAxis (PatchOptimizer.cpp:532-548), lane-parallel
            const DevView& RV = sc.views[row[op == OP_OBJ ? 0 : req_ref]];
            calc_patch_axes(RV.xaxis, RV.yaxis, n, scale, ax, ay, az);
        }
        BQ_PROF_END(tax_, 17);
        // ================= phase B: the wave serves all posted requests in one packed pass
        wave_lds_sync();  // orders the lanes' image-list updates (global workspace) before the cross-lane reads of phase B
        unsigned long long pending = __ballot(op != OP_NONE);
#ifdef HPMVS_PROFILE_PHASES
        { const long long t1 = clock64(); tA += t1 - t0; t0 = t1; nrounds++; nreq += __popcll(pending); }
#endif
        if (pending == 0ull) break;  // every lane idle and nothing posted: queue drained
        {
            PackedReq q;
            q.obj_mode = (op == OP_OBJ) ? 1 : 0;
            q.ref = req_ref; q.robust = req_robust; q.nimg = nimg; q.thr = req_thr;
            q.scale = scale;
            for (int k = 0; k < 3; k++) { q.c[k] = c[k]; q.zax[k] = (op == OP_OBJ) ? az[k] : n[k]; q.px[k] = ax[k]; q.py[k] = ay[k]; }
            PackedRes r;
            r.obj = 2.0; r.keep = 0ull; r.incc_sum = 0.0f; r.ngrabs = 0;
            wave_eval_rows<false>(sc, o, (const uint16_t*)row, pending, q, r, nullptr);
            if (op != OP_NONE) {
                res_f = r.obj;
                res_mask = r.keep;
                res_sum = r.incc_sum;
                ngrabs += r.ngrabs;
                op = OP_NONE;
            }
        }
        wave_lds_sync();
#ifdef HPMVS_PROFILE_PHASES
        { const long long t1 = clock64(); tB += t1 - t0; t0 = t1; }
#endif
    }
#ifdef HPMVS_PROFILE_PHASES
    unsigned long long* dbg = (unsigned long long*)(queue + 4);
    if (lane == 0) {
        atomicAdd(&dbg[0], (unsigned long long)tA);
        atomicAdd(&dbg[1], (unsigned long long)tB);
        atomicAdd(&dbg[2], (unsigned long long)nrounds);
        atomicAdd(&dbg[3], (unsigned long long)nreq);
    }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int k = 0; k < 64; k++) atomicAdd(&dbg[4 + k], s_prof[k]);
#endif
}

static int max_workgroups(int n_cus) {
    // workgroups per CU: LDS (optimiser state per wavefront, one staging buffer per workgroup) and the
    // waves-per-SIMD limit
    int lds_per_wg = (int)(sizeof(double) * BQ_ND * kP + sizeof(int) * BQ_NI * kP) + 16;
    lds_per_wg = (lds_per_wg + 511) & ~511;  // LDS allocation granule
    int per_cu = (160 * 1024) / lds_per_wg;
    if (per_cu > 4 * HPMVS_WAVES_PER_EU) per_cu = 4 * HPMVS_WAVES_PER_EU;
    return n_cus * per_cu;
}

size_t optimize_workspace_bytes(int n_cus) {
    return kCounterBytes + (size_t)max_workgroups(n_cus) * kP * kRow * sizeof(uint16_t);
}

void launch_optimize(const DevScene& sc, const DevOptions& o, const DevBatch& b, int32_t* workspace, int n_cus,
                     hipStream_t st) {
    if (b.n <= 0) return;
    // persistent single-wavefront workgroups, kP patches in flight each
    int wgs = b.n;  // up to one wavefront per patch (see `quota` in the kernel)
    const int cap = max_workgroups(n_cus);
    if (wgs > cap) wgs = cap;
    hipLaunchKernelGGL(optimize_kernel, dim3(wgs), dim3(64), 0, st, sc, o, b, workspace);
}

}  // namespace hpmvs
