// kernel_optimize.hip -- the batched PatchOptimizer::optimize kernel for gfx950.
//
// Mapping (MI355X-first, not a translation of the reference's one-thread-per-patch OpenMP loop):
//
//   * A persistent grid -- one workgroup of 8 wavefronts per CU -- pulls patches from a global work queue.
//   * Lanes 0..kP-1 of a wavefront each own one patch ("slot") and run that patch's control logic as a small
//     per-lane stage machine: the optimize() pipeline of the reference (src/hpmvs/PatchOptimizer.cpp:48-103:
//     addImages / filterImagesNCC / sortImages / assureImageAngles / optimizePatch / ... / setRefImage).
//     Whenever a lane needs something expensive it posts an op and stops; the wavefront serves the posted ops
//     cooperatively, 64 lanes wide:
//       - list services (pipeline.hpp): the image-list passes, one request at a time;
//       - optimiser service: bq_start / bq_step (the BOBYQA state machine, bobyqa3.hpp) for all lanes that have an
//         objective value to consume, with the patches' optimiser states in an LDS region taken from the CU's pool
//         (LDS TIME-SHARING below);
//       - photometric service ("phase B", wave_eval_rows in photometric.hpp): the (request, image) pairs become jobs,
//         one lane each prepares its image; then nine 7-lane groups (one lane per window ROW) each serve one request,
//         grab by grab, forming the reference's sequential float32 sums as DPP chains through the group's lanes.
//         No LDS, no barriers; the results go back to the requesting lanes.
//   * Patches finish at different times (40-1000 evaluations); a finished lane immediately pulls the next patch
//     from the queue, so slots stay busy until the queue drains.
//   * The image lists (uint16 ids, one 128-byte row per slot) and the parked optimiser states live in the launch
//     workspace (global memory, L2 resident); nothing per-patch goes through scratch memory.
//   * The file is compiled FOUR times (kernel_launch.hip chooses per launch):
//       this file                    optimize_kernel           29 slots per wavefront, 4 state regions per CU (all of a state in
//                                    + optimize_service_kernel  LDS): batches below 768 patches per CU; the open batch behind
//                                                               single-patch callers (2-wavefront workgroups, 1 region)
//       kernel_optimize_big.hip      optimize_kernel_big       64 slots (every lane owns a patch), 3 regions of 52 KB, 68 of a
//                                                               state's 164 float64 (bmat, zmat, vlag, wsuma) in global memory
//       kernel_optimize_tree.hip,    ..._tree, ..._big_tree    the same two with the opt-in throughput-mode window sums
//       kernel_optimize_big_tree.hip                            (HPMVS_SUMS=tree; not bit-identical to the oracle)
#include <hip/hip_runtime.h>
#include <atomic>

// optimiser state of the kP patches a wavefront owns lives in LDS while the wavefront is in phase A (bobyqa3.hpp)
#include "prof.h"
#define BQ_STATE_LDS 1
#define BQ_STATE_LDS_EXTERN 1
#ifndef BQ_P
#define BQ_P 29
#endif
// LDS TIME-SHARING.  The float64 optimiser blocks cost about the same per call whether 13, 29 or 64 lanes run them
// (measured per call: 109 k cycles at 29 lanes, 140 k at 64: profiles/r05/phase_profile_c2.txt), so patches per wavefront
// is THE lever -- and the 160 KB of LDS per CU, at 1364 B of optimiser state per patch, is what bounds it.  But a wavefront
// needs its optimiser state only inside the optimiser service; during the photometric service (no LDS) it is dead weight.
// So the 8 wavefronts of a CU share a POOL of state regions (four of 29 slots; three of 64 in the big build, which keeps the
// states' big matrices in global memory): the wavefront that enters the optimiser service takes a free region (LDS
// compare-and-swap locks), pages its states in from its parking buffer in global memory with LDS-DMA, runs bq_step
// kP lanes wide, pages out and releases.  Several times the lanes per optimiser block at the same occupancy.
#define BQ_SLOTS BQ_P
// The 64-slot build keeps the optimiser states' big matrices in global memory (kernel_optimize_big.hip sets this): a
// third state region per CU is worth far more there than the L2 round trips cost (C2, round 4: 232 -> 221 ms).  With 29
// slots (four regions of 40 KB already) it is the other way round (C1: 24.3 -> 25.5 ms), so this build keeps everything in LDS.
#ifndef HPMVS_STATIC_SLOTS
#define HPMVS_STATIC_SLOTS (!HPMVS_COLD_GLOBAL)   /* (the cold blocks are laid out per wavefront and slot: 29-slot build only) */
#endif
#ifndef HPMVS_STATIC_MAX_PER_WG
#define HPMVS_STATIC_MAX_PER_WG(W_, kS_) ((W_) * (kS_))
#endif
#ifndef HPMVS_SPREAD_SOLO
#define HPMVS_SPREAD_SOLO 1
#endif
#ifndef HPMVS_COLD_GLOBAL
#define HPMVS_COLD_GLOBAL 0
#endif
#if HPMVS_COLD_GLOBAL
#define BQ_COLD_GLOBAL 1     // the big arrays of a state stay in global memory (bobyqa3.hpp): from BQ_COLD_LO = 77 (bmat, zmat, vlag:
#ifndef BQ_COLD_LO           // 61 of the 164 float64 values) or 36 (xpt, fval, pq, hq as well: 102) up to index 138
#define BQ_COLD_LO 77
#endif
#ifndef BQ_COLD_HI
#define BQ_COLD_HI 138       // (145: wsuma as well)
#endif
#define HPMVS_STATE_ND (164 - (BQ_COLD_HI - BQ_COLD_LO))   // ... the others are what a state region holds per slot
#else
#define HPMVS_STATE_ND 164
#endif
#define HPMVS_STATE_NI 13
#define HPMVS_REGION_BYTES (HPMVS_STATE_ND * 8 * BQ_P + HPMVS_STATE_NI * 4 * BQ_P)
#define HPMVS_REGION_KIB ((HPMVS_REGION_BYTES + 1023) / 1024)  // paged in 1 KiB pieces (one LDS-DMA instruction each)
#define HPMVS_REGION_16 (HPMVS_REGION_KIB * 64)
// A batch launch runs ONE workgroup of 8 wavefronts per CU that share a POOL of 4 regions (a wavefront takes whichever
// is free): with a region per pair of wavefronts the partner was in the way 78 % of the time and a wavefront waited
// 18 k cycles per round for it; four servers behind one queue keep the same utilisation with about half the wait.
// The open batch (optimize_service_kernel) keeps two-wavefront workgroups with one region, so that it leaves three
// quarters of a CU's LDS to batch launches.  Regions and lock words are dynamic LDS: [regions][lock per region][done bits].
#ifndef HPMVS_WG_WAVES
#define HPMVS_WG_WAVES 8
#endif
#ifndef HPMVS_WG_REGIONS
#if HPMVS_COLD_GLOBAL
#define HPMVS_WG_REGIONS 6   // 29 slots x 876 B = 25 KiB per region
#else
#define HPMVS_WG_REGIONS 4
#endif
#endif
// This file is compiled four times (table at the top).  The externally visible names of the other copies carry
// HPMVS_VARIANT_SUFFIX, and only this copy has the open-batch kernel.
#ifdef HPMVS_VARIANT_SUFFIX
#define HPMVS_CAT2(a, b) a##b
#define HPMVS_CAT(a, b) HPMVS_CAT2(a, b)
#define HPMVS_V(name) HPMVS_CAT(name, HPMVS_VARIANT_SUFFIX)
#define HPMVS_VK(name) HPMVS_CAT(name, HPMVS_VARIANT_SUFFIX)
#else
#define HPMVS_VK(name) name  /* the standard kernel keeps its name (profiles, rocprof summaries) */
#define HPMVS_V(name) name##_std
#endif
extern __shared__ uint4 hpmvs_dyn[];
#define bq_lds_d ((double*)hpmvs_dyn)
#define bq_lds_i ((int*)((char*)hpmvs_dyn + HPMVS_STATE_ND * 8 * BQ_P))
#include "bobyqa3.hpp"
#include "launch.h"
#include "photometric.hpp"
#include "pipeline.hpp"

namespace hpmvs {

static_assert(BQ_ND_LDS == HPMVS_STATE_ND && BQ_NI == HPMVS_STATE_NI, "state region sized for another layout");

namespace {

constexpr int kP = BQ_P;  // patches ("slots") per wavefront: lanes 0..kP-1 own one each
constexpr int kRow = 64;  // uint16 ids per image-list row (global workspace, one row per slot)
constexpr size_t kCounterBytes = 1024;  // head of the launch workspace: work-queue counter (+ profiling counters)
constexpr int kPark16 = HPMVS_REGION_16;  // 16-byte chunks of a wavefront's parking buffer

#ifndef HPMVS_PARK_AUX
#define HPMVS_PARK_AUX 0
#endif
// PROXIES (round 5; VERDICT r4 #4: fewer SEQUENTIAL rounds per patch).  Two runs of evaluations of a patch do not depend on
// each other's values: prelim_'s seven start points (functions of the start point, the bounds and rhobeg: bq_prelim_points)
// and setRefImage's candidates (PatchOptimizer.cpp:154-181: one robust setINCCs per attached image).  A lane that is in one of
// them evaluates the next point itself, as before, and offers the FOLLOWING ones to lanes of its wavefront that own no patch
// in this round (the ramp-down of a launch, small batches, the open batch behind single-patch callers): the owner leaves what a
// request needs (centre, normal, scale, list) in its proxy row, a free lane posts the request, the photometric service
// serves it like any other, and the owner consumes the values in the reference's order -- bq_step once per value,
// `res_sum < refbest` in index order.  Same operations in the same order per patch; a patch needs up to 6 + (V - 1) fewer
// rounds.  Nothing changes while every lane owns a patch.  No call site and no second copy of set_center_norm is added
// to the persistent loop (either costs the hot path ~100 spilled registers: measured while building this).
// Measured (profiles/r05/proxies_ab.txt; all results byte-identical, the whole -m gpu suite green with HPMVS_PROXY=2): a
// host call of 8 patches 2.50 -> 2.32 ms, but the extra live values cost the batch kernels 2-4 % (C1 22.7 -> 23.7 ms, C2
// 215.3 -> 219.8 ms, 125 k seeds 37.1 -> 38.2 ms; 14-24 more spilled registers) for a ramp-down that gets no shorter (a patch
// that starts while the queue drains finds no free lane during its prelim_), and the open batch behind single-patch callers
// got SLOWER with them (2.50 -> 2.61 ms per call: 178 spilled registers against 135).  So the product is built WITHOUT
// proxies (HPMVS_PROXY = 0; 1: the open batch only, 2: every kernel); the code stays as an opt-in like HPMVS_FAST_SUMS.
#ifndef HPMVS_PROXY
#define HPMVS_PROXY 0
#endif
constexpr int kAuxDoubles = 64;  // a slot's proxy row: [0..17] prelim_'s start points 2..7, [18..24] the values this round's proxies
constexpr int kAuxVal = 18;      // delivered; as floats from kAuxCtx: scale, nimg, setRefImage's c / n, then c / n of the
constexpr int kAuxGrabs = 25;    // [25..31]: the grabs those evaluations took
constexpr int kAuxCtx = 64;      // start points 2..7
constexpr int kMaxExtra = 7;     // values a lane takes from proxies per round
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) float gfloat;
// the state region <-> this wavefront's parking buffer.  In: LDS-DMA (global_load_lds_dwordx4: 1 KiB per
// instruction straight into LDS at base + lane * 16, no staging registers, all ~39 pieces in flight at once).
// Out: ds_read_b128 + 16-byte global stores (fire and forget).
__device__ __forceinline__ void page_in(const uint4* __restrict__ park, uint4* hpmvs_region) {
    // One running per-lane pointer, four pieces per step through the instruction's immediate offset (which moves the
    // global AND the LDS address).  (With one precomputed address per piece the compiler hoisted all 39 out of the
    // kernel's main loop, spilled them, and reloaded each from scratch behind an s_waitcnt vmcnt(0) -- which also
    // waits for the previous piece: 39 serialised memory latencies, 27 k cycles per page-in.)
    const char* g = (const char*)(park + lane_id());
#pragma unroll
    for (int c4 = 0; c4 < HPMVS_REGION_KIB; c4 += 4) {
        asm volatile("" : "+v"(g));  // keep it a loop-carried value
        auto lds = (__attribute__((address_space(3))) void*)(hpmvs_region + c4 * 64);
        auto src = (const __attribute__((address_space(1))) void*)g;
        __builtin_amdgcn_global_load_lds(src, lds, 16, 0, HPMVS_PARK_AUX);
        if (c4 + 1 < HPMVS_REGION_KIB) __builtin_amdgcn_global_load_lds(src, lds, 16, 1024, HPMVS_PARK_AUX);
        if (c4 + 2 < HPMVS_REGION_KIB) __builtin_amdgcn_global_load_lds(src, lds, 16, 2048, HPMVS_PARK_AUX);
        if (c4 + 3 < HPMVS_REGION_KIB) __builtin_amdgcn_global_load_lds(src, lds, 16, 3072, HPMVS_PARK_AUX);
        g += 4096;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// HPMVS_PARK_NT_STORE=1 (lab): the parking stores carry the non-temporal hint (lines that come back a whole round later)
#ifndef HPMVS_PARK_NT_STORE
#define HPMVS_PARK_NT_STORE 0
#endif
#if HPMVS_PARK_NT_STORE
#define HPMVS_PARK_ST(dst, v) __builtin_nontemporal_store((v), &(dst))
#else
#define HPMVS_PARK_ST(dst, v) (dst) = (v)
#endif
// A REAL call: inlined into the kernel (every VGPR taken) the compiler left it one register quad, i.e. ds_read_b128,
// s_waitcnt, global_store, 78 times in a row -- 113 cycles per KiB, 8.8 k cycles per page-out with 58 slots, all of it
// while the region is held.  As a function it has the caller-saved registers to itself: eight reads in flight.
__device__ __attribute__((noinline)) void page_out(uint4* __restrict__ park, int region_index) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(1))) u4 gu4;
    const int lane = lane_id();
    const u4* region = (const u4*)(hpmvs_dyn + region_index * HPMVS_REGION_16);
    gu4* out = (gu4*)(unsigned long long)park;  // (global, not flat, stores)
    int c = 0;
#pragma unroll
    for (; c + 8 <= HPMVS_REGION_KIB; c += 8) {
        const u4 v0 = region[(c + 0) * 64 + lane], v1 = region[(c + 1) * 64 + lane], v2 = region[(c + 2) * 64 + lane],
                    v3 = region[(c + 3) * 64 + lane], v4 = region[(c + 4) * 64 + lane], v5 = region[(c + 5) * 64 + lane],
                    v6 = region[(c + 6) * 64 + lane], v7 = region[(c + 7) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);  // all eight reads are issued before the first store waits for its data
        HPMVS_PARK_ST(out[(c + 0) * 64 + lane], v0); HPMVS_PARK_ST(out[(c + 1) * 64 + lane], v1); HPMVS_PARK_ST(out[(c + 2) * 64 + lane], v2); HPMVS_PARK_ST(out[(c + 3) * 64 + lane], v3);
        HPMVS_PARK_ST(out[(c + 4) * 64 + lane], v4); HPMVS_PARK_ST(out[(c + 5) * 64 + lane], v5); HPMVS_PARK_ST(out[(c + 6) * 64 + lane], v6); HPMVS_PARK_ST(out[(c + 7) * 64 + lane], v7);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (; c < HPMVS_REGION_KIB; c++) { const u4 v = region[c * 64 + lane]; HPMVS_PARK_ST(out[c * 64 + lane], v); }
}

}  // namespace

// open batch: the records live in host memory that the host writes / reads while the kernel runs: cache-bypassing
// system-scope accesses, no cache maintenance
__device__ __forceinline__ unsigned long long ld_host64(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

#ifndef HPMVS_WAVES_PER_EU
#define HPMVS_WAVES_PER_EU 2
#endif
#ifndef HPMVS_BQ_PRIO
#define HPMVS_BQ_PRIO 3  // s_setprio while the state region is held (C2: 308 -> 302 ms)
#endif
#ifndef HPMVS_LOCK_SLEEP
#define HPMVS_LOCK_SLEEP 8
#endif
#ifndef HPMVS_A_PRIO
#define HPMVS_A_PRIO 2   // ... and in the rest of phase A (serial per-lane code the whole wavefront waits for); the
                        // photometric service runs at 0 (C2: 300 -> 296.5 ms)
#endif
// SERVICE = false: a batch launch (the arrays of `b` are device memory, n patches).  SERVICE = true: an open batch (see
// OPEN BATCH below).  Two instantiations, so that the batch kernel carries none of the open batch's code or registers.
template <bool SERVICE, int W, int R>
__device__ __forceinline__ void optimize_body(DevScene sc, DevOptions o, DevBatch b, int32_t* queue) {
    const int lane = lane_id();
    const int wv = (int)(threadIdx.x >> 6);            // which wavefront of the workgroup (W of them, R state regions)
    int* const hpmvs_lock = (int*)(hpmvs_dyn + R * HPMVS_REGION_16);  // [r]: 0 = region r free, 1 + w = held by wavefront w; [R]: bit w = wavefront w has left
    const int n_wgs = (int)gridDim.x;
    const int gw = wv * n_wgs + (int)blockIdx.x;        // wavefront index: the first wavefronts of all workgroups, then the second ones, ...
    // this wavefront's image-list rows and parking buffer in the launch workspace (behind the counter block)
    uint16_t* lists = (uint16_t*)((char*)queue + kCounterBytes) + (size_t)gw * kP * kRow;
    uint16_t* row = &lists[(lane < kP ? lane : 0) * kRow];
    uint4* park = (uint4*)((char*)queue + kCounterBytes + (size_t)W * n_wgs * kP * kRow * sizeof(uint16_t)) + (size_t)gw * kPark16;
    // ... and its result rows: per slot one row of kRow floats, where the photometric service leaves a request's per-image
    // values until the requesting lane adds them up (wave_eval_rows, step 7)
#if HPMVS_COLD_GLOBAL
    // ... and its block of the optimiser states' cold arrays (bobyqa3.hpp, BQ_COLD_GLOBAL): [index][slot], never paged
    char* cold = (char*)queue + kCounterBytes + (size_t)W * n_wgs * kP * kRow * sizeof(uint16_t) + (size_t)W * n_wgs * kPark16 * 16 +
                 (size_t)W * n_wgs * kP * kRow * sizeof(float) + (size_t)gw * BQ_NCOLD * kP * sizeof(double);
    {   // the same on all lanes: kept in scalar registers
        const unsigned long long a = (unsigned long long)cold;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(a & 0xffffffffull)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32));
        cold = (char*)(((unsigned long long)hi << 32) | lo);
    }
#endif
    const GlobalRows erows{(float*)((char*)queue + kCounterBytes + (size_t)W * n_wgs * kP * kRow * sizeof(uint16_t) +
                                    (size_t)W * n_wgs * kPark16 * 16) + (size_t)gw * kP * kRow, kRow};
    constexpr bool kProxy = HPMVS_PROXY == 2 || (HPMVS_PROXY == 1 && SERVICE);
    // ... and its slots' proxy rows (behind the cold blocks)
    gdouble* const auxw = (gdouble*)(unsigned long long)((char*)queue + kCounterBytes + (size_t)W * n_wgs * kP * kRow * sizeof(uint16_t) +
                          (size_t)W * n_wgs * kPark16 * 16 + (size_t)W * n_wgs * kP * kRow * sizeof(float) +
                          (size_t)W * n_wgs * (size_t)(HPMVS_COLD_GLOBAL ? (164 - HPMVS_STATE_ND) : 0) * kP * sizeof(double)) +
                          (size_t)gw * kP * kAuxDoubles;
    gdouble* const aux = auxw + (lane < kP ? lane : 0) * kAuxDoubles;  // my slot's row
    int pre_nf = 0;        // batched prelim_: the start point this lane is evaluating itself (1..7); 0: not in prelim_
    bool pre_ctx = false;  // ... and centre / normal of the start points after it are in my row
    int extra = 0;         // values my proxies delivered in this round (aux[kAuxVal ..])
    int px_owner = -1;     // this round I evaluate for that lane's patch ...
    int px_k = 0;          // ... its px_k-th extra point

    // ---- per-lane slot state
    // Small batches are spread over as many workgroups as possible (a wavefront with fewer busy slots has shorter
    // rounds: one lone patch advances ~2.4x faster than one of ten), and over the FIRST wavefronts of the
    // workgroups before any second one gets work: while a workgroup has no more busy wavefronts than regions, each keeps
    // its region and never pages.
    // OPEN BATCH (b.svc_ctrl != nullptr; the service behind single-patch host callers, capi.hip): the patch arrays
    // are host memory, the host appends patches while this launch runs, and a lane that holds queue ticket i waits
    // (without blocking its wavefront) until patch i has been published.  The first wavefront of workgroup 0 (the
    // dispatcher) copies the host's progress words into the launch workspace once per round; everybody else polls
    // that copy.  No cache maintenance anywhere in the protocol: host words, patch arrays and done flags are accessed
    // with cache-bypassing system-scope loads / stores, the copies with relaxed agent-scope atomics.
    constexpr bool service = SERVICE;
    const int per_wg = (b.n + n_wgs - 1) / n_wgs;
    int quota = per_wg - wv * kP;  // the workgroup's share goes to its wavefronts in order
    quota = quota < 0 ? 0 : (quota < kP ? quota : kP);
#if HPMVS_SPREAD_SOLO
    // ... unless it fits the wavefronts that can each keep a state region for good (no more busy wavefronts than regions:
    // `solo`, no paging): then it is split evenly between them -- four wavefronts with 4 patches each have much shorter rounds
    // than one with 16 (fewer photometric passes per round, and the optimiser's blocks run on four SIMDs at once).
    // Measured on the C1 scene, one launch (round 4): 1 024 patches 8.6 -> 7.0 ms, 4 096 10.9 -> 8.8, 10 000 12.8 -> 10.5,
    // 20 000 13.8 -> 13.0; at 29 000 (114 per workgroup) full wavefronts win again (16.1 against 17.1): up to 3/4 of R x kP.
    if (4 * per_wg <= 3 * R * kP) {
        const int a = per_wg < R ? per_wg : R;  // busy wavefronts of this workgroup
        quota = wv < a ? per_wg / a + (wv < per_wg % a ? 1 : 0) : 0;
    }
#endif
#if HPMVS_STATIC_SLOTS
    // ... and a share that fits W x kS slots goes to ALL wavefronts: W / R of them share a state region for good, each with
    // its own kS = kP / (W / R) slots of it (no lock, no paging: the regions are [index][slot] arrays, the wavefronts touch
    // disjoint slots)
    constexpr bool can_static = !SERVICE && (W % R == 0) && (W / R > 1);
    constexpr int kS = kP / (W / R > 0 ? W / R : 1);
    bool stat = false;
    if (can_static && per_wg <= HPMVS_STATIC_MAX_PER_WG(W, kS)) {
        stat = true;
        const int a = per_wg < W ? per_wg : W;
        quota = wv < a ? per_wg / a + (wv < per_wg % a ? 1 : 0) : 0;
    }
#endif
    if (service) quota = wv == 0 ? (b.svc_lanes < kP ? b.svc_lanes : kP) : 0;  // few patches per wavefront: latency first
    if (threadIdx.x <= R) hpmvs_lock[threadIdx.x] = 0;
    __syncthreads();
    if (quota == 0) {  // nothing to do for this wavefront: tell the others (fewer wavefronts than regions left: no more paging)
        if (lane == 0) atomicOr(&hpmvs_lock[R], 1 << wv);
        return;
    }
    __builtin_amdgcn_s_setprio(HPMVS_A_PRIO);
    bool held = false, solo = false;  // a region is held by this wavefront; no more wavefronts than regions are left (held for good, no paging)
    int reg = 0;                      // which region
    bool parked = false;              // my states are in the parking buffer
    int stage = (lane < quota) ? S_FETCH : S_IDLE;  // the other lanes only help in the wave-cooperative phases
    int idx = -1, nimg = 0, ii = 0, refindex = -1, ngrabs = 0, fail_stage = 0, nl_result = 0;
    float c0[3] = {0, 0, 0}, n0[3] = {0, 0, 0};  // optimize() input (kept for "patch untouched")
    float c[3] = {0, 0, 0}, n[3] = {0, 0, 1};    // pCenter_, pNormal_
    float scale = 1.0f, refbest = 0.0f;
    OptFields F;
    Bobyqa3 B;
    B.slot = lane < kP ? lane : 0;
    B.slot_i = B.slot;
#if HPMVS_STATIC_SLOTS
    if (stat) {
        reg = wv / (W / R);
        const int sl = (lane < kS ? lane : 0) + (wv % (W / R)) * kS;
        B.slot = sl + reg * (HPMVS_REGION_16 * 2);
        B.slot_i = sl + reg * (HPMVS_REGION_16 * 4);
        held = true; solo = true;
    }
#endif
#if HPMVS_COLD_GLOBAL
    B.coff = (unsigned)B.slot * 8u;
    B.coldp = cold;
    B.cold = __builtin_amdgcn_make_buffer_rsrc(cold, 0, BQ_NCOLD * BQ_CSTRIDE * 8, 0x00020000);  // (rebuilt as a scalar inside the optimiser's real functions)
#endif
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
    int svc_n = 0, svc_closed = 0;  // open batch: patches published so far / no more will come
    uint32_t* srec = service ? b.svc_stage + ((size_t)gw * kP + (lane < kP ? lane : 0)) * kSvcOutDwords : nullptr;
    int fin_idx = 0;
    bool fin_pending = false;  // open batch: my staging record waits for the wavefront to copy it to the host
    for (;;) {
        bool yield = false;  // open batch: my ticket's patch has not been published yet
        if (service) {
            if (gw == 0 && lane == 0) {  // the dispatcher: host progress words -> launch workspace
                const int closed = __hip_atomic_load(&b.svc_ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // closed before the count: a count read after it is final
                const int n_pub = __hip_atomic_load(&b.svc_ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&queue[200], n_pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the count is out before the closed flag
                __hip_atomic_store(&queue[201], closed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            int cl = 0, np = 0;
            if (lane == 0) {
                cl = __hip_atomic_load(&queue[201], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // closed is read first
                np = __hip_atomic_load(&queue[200], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            svc_closed = bcast_i(cl, 0);
            svc_n = bcast_i(np, 0);
        }
        // ================= phase A: every lane advances its own patch until it needs the wave
        while (op == OP_NONE && stage != S_IDLE && !yield) {
            bool finish_fail = false, finish_ok = false;
            switch (stage) {
            case S_FETCH:
                idx = atomicAdd(queue, 1);
                if (idx >= b.n) { stage = S_IDLE; break; }
                if (service) { stage = S_WAIT; break; }
                [[fallthrough]];
            case S_WAIT: {
                if (service && stage == S_WAIT) {
                    if (idx >= svc_n) {  // not published yet (or never: the batch was closed before my ticket)
                        if (svc_closed) stage = S_IDLE; else yield = true;
                        break;
                    }
                }
                bool bad_id = false;
                if (service) {  // my patch's input record (host memory): 4 + n_images / 2 eight-byte reads
                    const unsigned long long* in = b.svc_in + (size_t)idx * (kSvcInDwords / 2);
                    const unsigned long long w0 = ld_host64(in), w1 = ld_host64(in + 1), w2 = ld_host64(in + 2), w3 = ld_host64(in + 3);
                    c0[0] = __uint_as_float((unsigned)w0); c0[1] = __uint_as_float((unsigned)(w0 >> 32));
                    c0[2] = __uint_as_float((unsigned)w1); scale = __uint_as_float((unsigned)(w1 >> 32));
                    n0[0] = __uint_as_float((unsigned)w2); n0[1] = __uint_as_float((unsigned)(w2 >> 32));
                    n0[2] = __uint_as_float((unsigned)w3); nimg = (int)(unsigned)(w3 >> 32);
                    for (int k = 0; k < 3; k++) { c[k] = c0[k]; n[k] = n0[k]; }
                    ngrabs = 0; nl_result = 0;
                    srec[3] = 0;                                        // evaluations
                    for (int k = 20; k < 28; k++) srec[k] = 0;          // fmin, x
                    if (nimg < 0) { fail_stage = -nimg; nimg = 0; finish_fail = true; break; }
                    if (nimg > kMaxImages || nimg > b.max_images) { fail_stage = 100; finish_fail = true; break; }
                    for (int k = 0; k < nimg; k += 2) {
                        const unsigned long long w = ld_host64(in + 4 + k / 2);
                        const int id0 = (int)(unsigned)w, id1 = (int)(unsigned)(w >> 32);
                        bad_id |= (id0 < 0 || id0 >= sc.n_views);
                        row[k] = (uint16_t)id0;
                        if (k + 1 < nimg) { bad_id |= (id1 < 0 || id1 >= sc.n_views); row[k + 1] = (uint16_t)id1; }
                    }
                } else {
                    for (int k = 0; k < 3; k++) { c0[k] = b.center[4 * idx + k]; n0[k] = b.normal[4 * idx + k]; c[k] = c0[k]; n[k] = n0[k]; }
                    scale = b.scale[idx];
                    nimg = b.n_images[idx];
                    ngrabs = 0; nl_result = 0;
                    // optimiser diagnostics of a patch that never reaches the optimiser (overwritten when it finishes)
                    if (b.fmin) b.fmin[idx] = 0.0;
                    if (b.x) for (int k = 0; k < 3; k++) b.x[3 * idx + k] = 0.0;
                    if (b.nevals) b.nevals[idx] = 0;
                    if (nimg < 0) { fail_stage = -nimg; nimg = 0; finish_fail = true; break; }  // rejected by seed_init_kernel
                    if (nimg > kMaxImages || nimg > b.max_images) { fail_stage = 100; finish_fail = true; break; }
                    for (int k = 0; k < nimg; k++) {
                        const int id = b.images[(size_t)idx * b.max_images + k];
                        bad_id |= (id < 0 || id >= sc.n_views);
                        row[k] = (uint16_t)id;
                    }
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
                op = OPQ_START;  // bq_start in the optimiser service below
                stage = S_BQ;
                break;
            }
            case S_BQ:  // objective value arrived: advance BOBYQA (optimiser service below)
                op = OPQ_STEP;
                break;
            case S_BQFAIL:  // the optimiser ended with a code that optimizePatch treats as failure
                fail_stage = 4; finish_fail = true;
                break;
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
                if (kProxy) {   // what a proxy needs of me while I am in setRefImage
                    gfloat* ctx = (gfloat*)aux + kAuxCtx;
                    ctx[0] = scale; ctx[1] = __int_as_float(nimg);
                    for (int k = 0; k < 3; k++) { ctx[2 + k] = c[k]; ctx[5 + k] = n[k]; }
                }
                break;
            }
            case S_RS: {  // setRefImage (PatchOptimizer.cpp:154-181): one robust setINCCs per candidate
                if (res_sum < refbest) { refbest = res_sum; refindex = ii; }
                ii++;
                if (kProxy) {
                    for (int k = 0; k < extra; k++) {  // the candidates my proxies evaluated, in index order
                        const float sx = (float)aux[kAuxVal + k];
                        ngrabs += (int)aux[kAuxGrabs + k];
                        if (sx < refbest) { refbest = sx; refindex = ii; }
                        ii++;
                    }
                    extra = 0;
                }
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
                if (service) {
                    // my slot's staging record (device memory); the wavefront copies it to the host below
                    srec[0] = finish_ok ? 1u : 0u; srec[1] = (unsigned)(finish_ok ? 0 : fail_stage); srec[2] = (unsigned)nl_result;
                    srec[4] = (unsigned)ngrabs; srec[5] = (unsigned)(finish_ok ? nimg : 0);
                    if (finish_ok) {
                        for (int k = 0; k < 3; k++) { srec[8 + k] = __float_as_uint(c[k]); srec[12 + k] = __float_as_uint(n[k]); srec[16 + k] = __float_as_uint(res_rgb[k]); }
                        srec[11] = __float_as_uint(1.0f); srec[15] = __float_as_uint(0.0f);
                        srec[19] = __float_as_uint(1.4f);  // PatchOptimizer.cpp:95
                        for (int k = 0; k < nimg && k < b.max_images; k++) srec[32 + k] = row[k];
                    }
                    fin_idx = idx; fin_pending = true; yield = true;
                } else {
                    // (2: the list outgrew this kernel's 64-id rows -- optimize_wide_kernel, launched behind this one, redoes the
                    //  patch from its untouched inputs: kernel_wide.hip)
                    b.ok[idx] = finish_ok ? 1 : (fail_stage == 100 ? 2 : 0);
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
                    if (b.result) b.result[idx] = nl_result;
                    if (b.ngrabs) b.ngrabs[idx] = ngrabs;
                }
                stage = S_FETCH;
            }
        }
        // ================= open batch: finished patches go to the host, one 384-byte record each, written by the whole
        // wavefront (two coalesced stores + one flag: host-memory transactions are what bounds the callers' rate)
        if (service) {
            const unsigned long long fm = __ballot(fin_pending);
            if (fm) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the lanes' staging stores are out
                for (unsigned long long t = fm; t; t &= t - 1ull) {
                    const int p = __ffsll((long long)t) - 1;
                    const int pidx = bcast_i(fin_idx, p);
                    const uint32_t* src = b.svc_stage + ((size_t)gw * kP + p) * kSvcOutDwords;
                    uint32_t* dst = b.svc_out + (size_t)pidx * kSvcOutDwords;
                    const uint32_t v0 = __hip_atomic_load(src + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t v1 = lane < kSvcOutDwords - 64 ? __hip_atomic_load(src + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                    __hip_atomic_store(dst + lane, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (lane < kSvcOutDwords - 64) __hip_atomic_store(dst + 64 + lane, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // records first, then the flags the owners spin on
                if (fin_pending) __hip_atomic_store(&b.svc_done[fin_idx], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                fin_pending = false;
            }
        }
        // ================= list services: the image-list passes, one request at a time, 64 lanes wide (pipeline.hpp)
        {
            wave_lds_sync();  // the lanes' own list writes (global workspace) before the cooperative reads
            const unsigned long long lm = __ballot(op >= OPL_ADD && op <= OPL_F3);
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
                    const ListEnv le = list_env(sc, o);
                    if (kind == OPL_ADD) r.nimg = coop_add_images(le, prow, pn, pc, pnrm, pscale);
                    else if (kind == OPL_F1) r = coop_list_f1(le, prow, pn, pmask, pc, pnrm, pscale);
                    else if (kind == OPL_F2) r = coop_list_f2(le, prow, pn, pmask, pc, pnrm);
                    else { LevelThr lt; for (int k = 0; k < kMaxLevels; k++) lt.t[k] = sc.lvl_thr[k]; r = coop_list_f3(sc.views, lt, o.MIN_IMAGES_PER_PATCH, prow, pn, pmask, pc, pscale, b.max_images, service || b.color != nullptr); }
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
        // ================= optimiser service: the lanes whose patch needs bq_start / bq_step run them together,
        // holding the workgroup's state region only for that long (see LDS TIME-SHARING above)
        if (__any(op >= OPQ_START)) {
            const bool mine = op >= OPQ_START;
            double x0[3] = {0, 0, 0}, dx[3] = {1, 1, 1};
            if (op == OPQ_START) {
                parameters_from_center_norm(F, o, n, lbv, ubv, x0);
                bq_default_step(x0, lbv, ubv, dx);
            }
            if (!held) {  // take the region and bring my patches' states in
                BQ_PROF_BEGIN(tpg_);
                int got = 0;
                if (lane == 0) {
                    int r = R > 1 ? wv % R : 0;  // (start at different regions: fewer failed attempts)
                    for (;;) {
                        if (atomicCAS(&hpmvs_lock[r], 0, 1 + wv) == 0) break;
                        r = r + 1 == R ? 0 : r + 1;
                        if (r == (R > 1 ? wv % R : 0)) __builtin_amdgcn_s_sleep(HPMVS_LOCK_SLEEP);  // every region is taken
                    }
                    got = r;
                }
                reg = __builtin_amdgcn_readfirstlane(got);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                BQ_PROF_END(tpg_, 25);
                // my patches' states sit in region `reg` from here on: offsets of its arrays in 8- and 4-byte words
                B.slot = (lane < kP ? lane : 0) + reg * (HPMVS_REGION_16 * 2);
                B.slot_i = (lane < kP ? lane : 0) + reg * (HPMVS_REGION_16 * 4);
                BQ_PROF_BEGIN(tpi_);
                if (parked) page_in(park, hpmvs_dyn + reg * HPMVS_REGION_16);  // (nothing is parked before this wavefront's first release)
                BQ_PROF_END(tpi_, 27);
                held = true;
                // wave-uniform LDS read: when no more wavefronts than regions are left, nobody waits for mine any more
                solo = W - __popc((unsigned)hpmvs_lock[R] & ((1u << W) - 1u)) <= R;
            }
#if HPMVS_BQ_PRIO
            __builtin_amdgcn_s_setprio(HPMVS_BQ_PRIO);  // the region is the contended resource: hold it as briefly as possible
#endif
            int ret = BQ_DONE;
#ifdef HPMVS_HOLD_DELAY  /* experiment: how sensitive is the launch to the time a state region is held? */
            for (int dly = 0; dly < HPMVS_HOLD_DELAY; dly++) __builtin_amdgcn_s_sleep(127);
#endif
            if (kProxy) {
                if (op == OPQ_START) {
                    ret = bq_start(B, x0, lbv, ubv, dx, 1.e-7, 1000, (double*)(unsigned long long)aux);  // (+ prelim_'s start points 2..7 into my row)
                    pre_nf = ret == BQ_CONTINUE ? 1 : 0; pre_ctx = false;
                }
                // bq_step with my own value; in prelim_ then with the values my proxies delivered, in order (ONE call site)
                const bool pre = op == OPQ_STEP && pre_nf > 0;
                bool stepping = op == OPQ_STEP;
                double fcur = res_f;
                int kx = 0;
                for (;;) {
                    if (stepping) { ret = bq_step(B, fcur); pre_nf += pre ? 1 : 0; }
                    stepping = pre && kx < extra && ret == BQ_CONTINUE;
                    if (!__any(stepping)) break;
                    if (stepping) { fcur = (double)aux[kAuxVal + kx]; ngrabs += (int)aux[kAuxGrabs + kx]; kx++; }
                }
                if (pre) { if (pre_nf > 7 || ret != BQ_CONTINUE) pre_nf = 0; extra = 0; }  // (> 7: prelim_ is over)
            } else {
                if (op == OPQ_START) ret = bq_start(B, x0, lbv, ubv, dx, 1.e-7, 1000);
                if (op == OPQ_STEP) ret = bq_step(B, res_f);
            }
            double xe[3] = {0, 0, 0};
            if (mine) {
                for (int k = 0; k < 3; k++) xe[k] = BQA(xeval, k);  // next point, or the final one (same slot)
                if (ret != BQ_CONTINUE) {
                    nl_result = BQN(rc);
                    if (service) {
                        const unsigned long long mf = (unsigned long long)__double_as_longlong((double)BQS(minf));
                        srec[3] = (unsigned)BQN(nevals);
                        srec[20] = (unsigned)mf; srec[21] = (unsigned)(mf >> 32);
                        for (int k = 0; k < 3; k++) {
                            const unsigned long long xv = (unsigned long long)__double_as_longlong(xe[k]);
                            srec[22 + 2 * k] = (unsigned)xv; srec[23 + 2 * k] = (unsigned)(xv >> 32);
                        }
                    } else {
                        if (b.fmin) b.fmin[idx] = BQS(minf);
                        if (b.x) for (int k = 0; k < 3; k++) b.x[3 * idx + k] = xe[k];
                        if (b.nevals) b.nevals[idx] = BQN(nevals);
                    }
                }
            }
            if (!solo) {  // park my states and give the region back to the pool
                BQ_PROF_BEGIN(tpo_);
                page_out(park, reg);
                parked = true;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) atomicExch(&hpmvs_lock[reg], 0);
                held = false;
                BQ_PROF_END(tpo_, 26);
            }
#if HPMVS_BQ_PRIO
            __builtin_amdgcn_s_setprio(HPMVS_A_PRIO);
#endif
            {
                const bool cont = mine && ret == BQ_CONTINUE;
                // bq_start refused the arguments, or a result other than SUCCESS / STOPVAL / FTOL / XTOL (PatchOptimizer.cpp:367-368; negative codes throw)
                const bool failed = mine && !cont && (op == OPQ_START || !(nl_result >= 1 && nl_result <= 4));
                const bool done = mine && !cont && !failed;
                int nset = (cont || done) ? 1 : 0;  // setCenterNorm (PatchOptimizer.cpp:401-414) of the next point / the final one ...
                // ... and, once per patch and only while lanes are free to act as proxies, of prelim_'s start points after it
                const int pfirst = pre_nf;  // the start point `xe` is
                if (kProxy) {
                    const bool anyfree = __any(!mine && op == OP_NONE && lane < kP && (stage == S_IDLE || (service && stage == S_WAIT)));
                    if (anyfree && cont && pre_nf >= 1 && pre_nf <= 6 && !pre_ctx) { nset = 8 - pre_nf; pre_ctx = true; }
                }
                for (int k = 0; __any(k < nset); k++) {  // (ONE inlined copy of set_center_norm in the persistent loop)
                    if (k < nset) {
                        double xk[3] = {xe[0], xe[1], xe[2]};
                        float ck[3], nk[3];
                        if (kProxy && k > 0) for (int i3 = 0; i3 < 3; i3++) xk[i3] = aux[3 * (pfirst + k - 2) + i3];  // start point pfirst + k
                        set_center_norm(F, o, xk, ck, nk);
                        if (k == 0) { for (int i3 = 0; i3 < 3; i3++) { c[i3] = ck[i3]; n[i3] = nk[i3]; } }
                        else if (kProxy) {
                            gfloat* ctx = (gfloat*)aux + kAuxCtx + 8 + 6 * (pfirst + k - 2);
                            for (int i3 = 0; i3 < 3; i3++) { ctx[i3] = ck[i3]; ctx[3 + i3] = nk[i3]; }
                            gfloat* c0 = (gfloat*)aux + kAuxCtx;
                            c0[0] = scale; c0[1] = __int_as_float(nimg);
                        }
                    }
                }
                if (cont) op = OP_OBJ;
                else if (failed) { op = OP_NONE; stage = S_BQFAIL; }
                else if (done) { op = OPL_ADD; stage = S_ADD2; }  // addImages
            }
            // failed lanes finish in phase A; finished optimisations go on with addImages in the list services
            if (__any((op == OP_NONE && stage != S_IDLE) || (op >= OPL_ADD && op <= OPL_F3))) continue;
        }
        const uint16_t* brow = row;  // the image list my request of this round refers to (a proxy's: its owner's)
        if (kProxy) {
            // lanes that own no patch in this round ...
            unsigned long long freem = __ballot(op == OP_NONE && lane < kP && (stage == S_IDLE || (service && stage == S_WAIT)) && !fin_pending);
            // ... and lanes with further independent evaluations: prelim_'s start points pre_nf + 1 .. 7, setRefImage's candidates req_ref + 1 .. nimg - 1
            int want = 0;
            if (op == OP_OBJ && pre_ctx && pre_nf >= 1 && pre_nf <= 6) want = 7 - pre_nf;
            else if (op == OP_REFSUM && stage == S_RS) want = nimg - 1 - req_ref;
            want = want > kMaxExtra ? kMaxExtra : want;
            unsigned long long own = freem ? __ballot(want > 0) : 0ull;
            if (own) {  // (wave-uniform)
                for (; own && freem; own &= own - 1ull) {
                    const int p = __ffsll((long long)own) - 1;
                    const int w = bcast_i(want, p);
                    int cnt = 0;
                    for (; cnt < w && freem; cnt++, freem &= freem - 1ull)
                        if (lane == __ffsll((long long)freem) - 1) { px_owner = p; px_k = cnt; }
                    if (lane == p) extra = cnt;
                }
                // a proxy's request, from its owner's row (the proxy's own patch variables are dead: it owns no patch)
                const int src = px_owner >= 0 ? px_owner : lane;
                const int o_op = shfl_i(op, src), o_ref = shfl_i(req_ref, src), o_pre = shfl_i(pre_nf, src);
                if (px_owner >= 0) {
                    brow = lists + px_owner * kRow;
                    const gfloat* ctx = (const gfloat*)(auxw + px_owner * kAuxDoubles) + kAuxCtx;
                    scale = ctx[0]; nimg = __float_as_int(ctx[1]);
                    const gfloat* cn = (o_op == OP_OBJ) ? ctx + 8 + 6 * (o_pre - 1 + px_k) : ctx + 2;  // start point o_pre + 1 + px_k; or setRefImage's patch
                    for (int k = 0; k < 3; k++) { c[k] = cn[k]; n[k] = cn[3 + k]; }
                    if (o_op == OP_OBJ) op = OP_OBJ;
                    else { op = OP_REFSUM; req_ref = o_ref + 1 + px_k; req_robust = 1; }  // candidate o_ref + 1 + px_k of setRefImage
                }
            }
        }
        BQ_PROF_BEGIN(tax_);
        if (op != OP_NONE) {  // calculatePatchAxis (PatchOptimizer.cpp:532-548), lane-parallel
            const DevView& RV = sc.views[brow[op == OP_OBJ ? 0 : req_ref]];
            calc_patch_axes(RV.xaxis, RV.yaxis, n, scale, ax, ay, az);
        }
        BQ_PROF_END(tax_, 17);
        // ================= phase B: the wave serves all posted requests in one packed pass
        wave_lds_sync();  // orders the lanes' image-list updates (global workspace) before the cross-lane reads of phase B
        unsigned long long pending = __ballot(op != OP_NONE);
#ifdef HPMVS_PROFILE_PHASES
        { const long long t1 = clock64(); tA += t1 - t0; t0 = t1; nrounds++; nreq += __popcll(pending); }
#endif
        if (pending == 0ull) {  // nothing posted
            // open batch: lanes are waiting for their patch; and the dispatcher stays until the batch is closed
            if (service && (__any(stage != S_IDLE) || (gw == 0 && !svc_closed))) {
                // ~10 us between polls: a waiting wavefront must not take issue slots or cache contents from the ones
                // that work (1024 waiting workgroups polling every microsecond made a lone patch 5x slower); the
                // dispatcher reads the host words every ~1 us
                if (gw == 0) __builtin_amdgcn_s_sleep(32);
                else { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); }
                continue;
            }
            if (lane == 0) atomicOr(&hpmvs_lock[R], 1 << wv);  // every lane idle: queue drained
            break;
        }
        {
            PackedReq q;
            q.obj_mode = (op == OP_OBJ) ? 1 : 0;
            q.ref = req_ref; q.robust = req_robust; q.nimg = nimg; q.thr = req_thr;
            q.scale = scale;
            for (int k = 0; k < 3; k++) { q.c[k] = c[k]; q.zax[k] = (op == OP_OBJ) ? az[k] : n[k]; q.px[k] = ax[k]; q.py[k] = ay[k]; }
            PackedRes r;
            r.obj = 2.0; r.keep = 0ull; r.incc_sum = 0.0f; r.ngrabs = 0;
            __builtin_amdgcn_s_setprio(0);
            wave_eval_rows<false>(sc, o, brow, pending, q, r, erows);
            __builtin_amdgcn_s_setprio(HPMVS_A_PRIO);
            if (op != OP_NONE) {
                res_f = r.obj;
                res_mask = r.keep;
                res_sum = r.incc_sum;
                ngrabs += r.ngrabs;
                op = OP_NONE;
            }
            if (kProxy) {   // what the proxies of this round found goes to their owners' rows (value k of an owner: its k-th extra point)
                const unsigned long long pm = __ballot(px_owner >= 0);
                if (pm) {
                    for (unsigned long long t = pm; t; t &= t - 1ull) {
                        const int fl = __ffsll((long long)t) - 1;
                        const int p = bcast_i(px_owner, fl), k = bcast_i(px_k, fl), g = bcast_i(r.ngrabs, fl), kind = bcast_i(q.obj_mode, fl);
                        const double vf = bcast_d(res_f, fl);
                        const float vs = bcast_f(res_sum, fl);
                        // (the proxy's grabs wait beside its value and are counted when the owner CONSUMES the value -- a value that is
                        //  dropped because bq_step stopped inside prelim_ was never part of the reference's run: ADVICE r5)
                        if (lane == p) { aux[kAuxVal + k] = kind ? vf : (double)vs; aux[kAuxGrabs + k] = (double)g; }
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the owners read these values in their next phase A
                    px_owner = -1;
                }
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

__global__ void __launch_bounds__(64 * HPMVS_WG_WAVES, HPMVS_WAVES_PER_EU) HPMVS_VK(optimize_kernel)(DevScene sc, DevOptions o, DevBatch b, int32_t* queue) {
    optimize_body<false, HPMVS_WG_WAVES, HPMVS_WG_REGIONS>(sc, o, b, queue);
}
#ifndef HPMVS_VARIANT_SUFFIX
__global__ void __launch_bounds__(128, HPMVS_WAVES_PER_EU) optimize_service_kernel(DevScene sc, DevOptions o, DevBatch b, int32_t* queue) {
    optimize_body<true, 2, 1>(sc, o, b, queue);
}
#endif

static size_t lds_bytes(int regions) { return (size_t)regions * HPMVS_REGION_16 * 16 + (size_t)(regions + 1) * sizeof(int); }

static int max_workgroups(int n_cus, int waves, int regions) {
    // workgroups per CU: LDS (their state regions) and the waves-per-SIMD limit
    int lds_per_wg = (int)lds_bytes(regions);
#ifdef HPMVS_PROFILE_PHASES
    lds_per_wg += 64 * 8;
#endif
    lds_per_wg = (lds_per_wg + 511) & ~511;  // LDS allocation granule
    int per_cu = (160 * 1024) / lds_per_wg;
    if (per_cu > 4 * HPMVS_WAVES_PER_EU / waves) per_cu = 4 * HPMVS_WAVES_PER_EU / waves;
    return n_cus * per_cu;
}

size_t HPMVS_V(optimize_workspace_bytes)(int n_cus) {
    // counter block + per wavefront: kP image-list rows, one parking buffer for the optimiser states, kP result rows
    const size_t waves = (size_t)HPMVS_WG_WAVES * max_workgroups(n_cus, HPMVS_WG_WAVES, HPMVS_WG_REGIONS);
    return kCounterBytes + waves * kP * kRow * sizeof(uint16_t) + waves * (size_t)kPark16 * 16 + waves * kP * kRow * sizeof(float) +
           waves * (size_t)(HPMVS_COLD_GLOBAL ? (164 - HPMVS_STATE_ND) : 0) * kP * sizeof(double) +
           (HPMVS_PROXY == 2 ? waves * kP * kAuxDoubles * sizeof(double) : 0);
}

#ifndef HPMVS_VARIANT_SUFFIX
// open batch (b.svc_ctrl set): a fixed grid of `wgs` two-wavefront workgroups (one per CU leaves room for batch launches beside it)
size_t optimize_workspace_bytes_for(int wgs) {
    return kCounterBytes + (size_t)2 * wgs * kP * kRow * sizeof(uint16_t) + (size_t)2 * wgs * (size_t)kPark16 * 16 +
           (size_t)2 * wgs * kP * kRow * sizeof(float) + (size_t)2 * wgs * (size_t)(HPMVS_COLD_GLOBAL ? (164 - HPMVS_STATE_ND) : 0) * kP * sizeof(double) +
           (HPMVS_PROXY ? (size_t)2 * wgs * kP * kAuxDoubles * sizeof(double) : 0);
}
size_t optimize_stage_bytes_for(int wgs) { return (size_t)2 * wgs * kP * kSvcOutDwords * sizeof(uint32_t); }
void launch_optimize_service(const DevScene& sc, const DevOptions& o, const DevBatch& b, int32_t* workspace, int wgs,
                             hipStream_t st) {
    hipLaunchKernelGGL(optimize_service_kernel, dim3(wgs), dim3(128), lds_bytes(1), st, sc, o, b, workspace);
}
#endif

void HPMVS_V(launch_optimize)(const DevScene& sc, const DevOptions& o, const DevBatch& b, int32_t* workspace, int n_cus,
                     hipStream_t st) {
    if (b.n <= 0) return;
    // persistent workgroups, up to kP patches in flight per wavefront
    int wgs = b.n;  // up to one workgroup per patch (see `quota` in the kernel)
    const int cap = max_workgroups(n_cus, HPMVS_WG_WAVES, HPMVS_WG_REGIONS);
    if (wgs > cap) wgs = cap;
    // more than 64 KB of dynamic LDS needs the attribute -- once per device of the process
    static std::atomic<unsigned long long> attr_done{0ull};
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && !((attr_done.load(std::memory_order_relaxed) >> dev) & 1ull)) {
        (void)hipFuncSetAttribute((const void*)HPMVS_VK(optimize_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(HPMVS_WG_REGIONS));
        attr_done.fetch_or(1ull << dev, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(HPMVS_VK(optimize_kernel), dim3(wgs), dim3(64 * HPMVS_WG_WAVES), lds_bytes(HPMVS_WG_REGIONS), st, sc, o, b, workspace);
}

}  // namespace hpmvs
