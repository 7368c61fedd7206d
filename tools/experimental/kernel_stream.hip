// kernel_stream.hip -- the throughput form of the batched PatchOptimizer::optimize kernel for gfx950.
//
// kernel_optimize.hip binds every patch to one lane of one wavefront for its whole life.  The lanes of a
// wavefront then sit in DIFFERENT blocks of the optimiser, every block (trsbox, altmov, update, ...) runs in
// (almost) every round for the one or two lanes that need it, and the per-CU patch count is capped by LDS: the
// float64 trust-region algebra executes with ~10 of 64 lanes for ~2/3 of the kernel's cycles.
//
// Here patches are NOT bound to lanes.  The state of every in-flight patch (optimiser state, pipeline fields,
// image list: one 1792-byte record) lives in HBM / MALL, and patches STREAM through per-block work queues:
//
//     FETCH -> EVAL -> PIPE -> EVAL -> AE -> L60 -> L230 -> EVAL -> AE -> ... -> PIPE -> EVAL -> ... -> done
//
//   * a persistent wavefront pops up to SK_P patches that wait at the SAME block, pages their optimiser state
//     into LDS ([index][lane], the layout bobyqa3.hpp already uses), runs that block for all of them at once,
//     lets the cheap transition labels (L20/L650/L680/L360/L720) run inline, pages the state back and pushes
//     every patch to the queue of its next block;
//   * photometric requests (objective_fn, setINCCs) of 64 DIFFERENT patches are served by one EVAL visit;
//   * the image-list passes of runOptimization (addImages, filterImagesNCC, sortImages, ...) are PIPE visits.
//
// Every float32 / float64 operation is the one kernel_optimize.hip (and the reference) executes: the blocks are
// the same functions of bobyqa3.hpp / photometric.hpp / pipeline.hpp; only WHICH wavefront runs WHICH block for
// WHICH patch changes.  Results are bit-identical to kernel_optimize.hip and to the oracle.
//
// Inter-wavefront hand-off (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup
// visibility"): per-XCD L2s are not coherent and a CU's L1 is never refreshed by other CUs' stores, so every
// record word is written with an sc1 (write-through) store and read with an sc1 (L1-bypassing) load
// (8-byte relaxed agent-scope atomics), the writing wavefront drains its stores (s_waitcnt vmcnt(0)) before it
// publishes the patch in a queue, and queue entries carry a lap tag so that a slot is never read before it was
// written.  No fences, no dispatch-order or placement assumption.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "prof.h"
#define BQ_STATE_LDS 1
#define BQ_STATE_LDS_EXTERN 1
#ifndef SK_P
#define SK_P 13  // patches per block visit (= lanes running float64 algebra); SK_P * 1468 B of LDS per wavefront
#endif
#define BQ_P SK_P
#define BQ_SLOTS SK_P

namespace hpmvs {
constexpr int kSkStateBytes = (8 * 177 + 4 * 13) * SK_P;
constexpr int kSkEvalBytes = 5632;  // >= sizeof(EvalLds), checked below
constexpr int kSkLdsBytes = ((kSkStateBytes > kSkEvalBytes ? kSkStateBytes : kSkEvalBytes) + 15) & ~15;
// one LDS buffer per wavefront: optimiser state of the visit's patches, or the texel staging of an EVAL visit
__shared__ __attribute__((aligned(16))) unsigned char sk_lds[kSkLdsBytes];
}  // namespace hpmvs
#define bq_lds_d ((double*)hpmvs::sk_lds)
#define bq_lds_i ((int*)(hpmvs::sk_lds + 8 * 177 * SK_P))
#include "bobyqa3.hpp"
#include "launch.h"
#include "photometric.hpp"
#include "pipeline.hpp"

namespace hpmvs {

static_assert(BQ_ND == 177 && BQ_NI == 13, "sk_lds layout assumes 177 doubles + 13 ints per patch");
static_assert(sizeof(EvalLds) <= kSkEvalBytes, "EVAL staging does not fit the shared buffer");

namespace {

typedef unsigned long long u64;

// ---- coherent (sc1) word access -------------------------------------------------------------------------
__device__ __forceinline__ u64 ldc(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stc(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ldc32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stc32(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

#ifdef SK_STATS
#define SKV_T(v) const long long v = clock64()
#define SKV_ACC(ws_, i, a, b_) do { if (lane_id() == 0) atomicAdd((ws_).ctl + kCtlStats + (i), (u64)((b_) - (a))); } while (0)
#else
#define SKV_T(v) do { } while (0)
#define SKV_ACC(ws_, i, a, b_) do { } while (0)
#endif

// ---- record layout (u64 words) ----------------------------------------------------------------------------
// 2048 bytes = 8 pages of 32 words (256 B).  The optimiser state is stored in the order the blocks need it, so
// that a visit pages in a PREFIX of the record and pages out a few whole pages:
//   page 0-1  words   0..45  small vectors xeval..gnew (state 0..44) + pad
//             words  46..69  scalars minf..stepb (state 154..176) + pad
//   page 2    words  70..77  the 13 int32 + pad
//             words  78..105 xpt, fval (state 45..72)
//   page 3    words 106..119 pq, hq (state 73..85) + pad          <- L60 (trsbox) reads pages 0..3
//   page 3-4  words 120..149 bmat (state 86..115)
//   page 4-5  words 150..171 zmat (state 116..136) + pad          <- L90 / L210 / L230 read pages 0..5
//   page 5    words 172..181 vlag (state 137..146); 182..189 wsuma (state 147..153) + pad; 190, 191 pad
//   page 6    words 192..215 PipeState; 216..231 image list (64 uint16); 232..255 unused
constexpr int kRecPS = 192;
constexpr int kRecRow = 216;
constexpr int kRecWords = 256;
constexpr int kRowWords = 16;
constexpr int kPageWords = 32;
// record word of a float64 state index (bobyqa3.hpp O_*) / of int32 state index k (two per word)
__host__ __device__ constexpr int rec_word_of_d(int i) {
    return i < 45 ? i : i < 73 ? 78 + (i - 45) : i < 86 ? 106 + (i - 73) : i < 116 ? 120 + (i - 86) : i < 137 ? 150 + (i - 116)
         : i < 147 ? 172 + (i - 137) : i < 154 ? 182 + (i - 147) : 46 + (i - 154);
}
__host__ __device__ constexpr int rec_word_of_i(int k) { return 70 + k / 2; }
// inverse: state index stored in record word w (-1: pad / not a float64 word)
__host__ __device__ constexpr int d_of_rec_word(int w) {
    return w < 45 ? w : w == 45 ? -1 : w < 69 ? 154 + (w - 46) : w < 78 ? -1 : w < 106 ? 45 + (w - 78) : w < 119 ? 73 + (w - 106)
         : w == 119 ? -1 : w < 150 ? 86 + (w - 120) : w < 171 ? 116 + (w - 150) : w == 171 ? -1 : w < 182 ? 137 + (w - 172)
         : w < 189 ? 147 + (w - 182) : -1;
}
// pages a visit reads (a prefix) and writes (bit mask)
constexpr int kPagesAll = 6;

// stages of runOptimization a patch can wait in (PatchOptimizer.cpp:48-76), see the PIPE visit
enum { T_F1 = 1, T_BQ = 2, T_BQDONE = 3, T_F2 = 4, T_RS = 5, T_F3 = 6 };

struct alignas(8) PipeState {
    int idx, nimg, stage, ii;
    int refindex, ngrabs, fail_stage, nl_result;
    int op, req_ref, req_robust, have_bq;
    float req_thr, scale, refbest, res_sum;
    float c0[3], n0[3], c[3], n[3];
    float F[15];   // OptFields: refCenter, refRay, imgX, imgY, imgZ
    float pad;
    double res_f;
    u64 res_mask;
};
static_assert(sizeof(PipeState) == 24 * 8, "PipeState is 24 record words");
static_assert(sizeof(OptFields) == 15 * 4, "OptFields is 15 floats");

// ---- queues ------------------------------------------------------------------------------------------------
enum { Q_EVAL = 0, Q_AE, Q_L60, Q_L90, Q_L210, Q_L230, Q_RESC, Q_PIPE, Q_FREE, Q_COUNT };
// control block (u64 words, zeroed per launch): queue q has its head at word 16q and its tail at 16q + 8 (own
// 64-byte lines); scalars behind the queues
constexpr int kCtlNextIdx = 16 * Q_COUNT;       // next input patch
constexpr int kCtlNextSlot = kCtlNextIdx + 8;   // next never-used record
constexpr int kCtlDone = kCtlNextSlot + 8;      // finished patches
constexpr int kCtlAbort = kCtlDone + 8;         // watchdog / internal-error flag
constexpr int kCtlStats = kCtlAbort + 8;         // SK_STATS builds: 64 counters
constexpr int kCtlWords = kCtlStats + 80;
constexpr int kSlotBits = 18;                   // records per launch <= 262144
constexpr unsigned kSlotMask = (1u << kSlotBits) - 1u;
constexpr unsigned kTagMask = (1u << (32 - kSlotBits)) - 1u;

struct SkWs {
    u64* ctl;
    unsigned* rings;   // Q_COUNT rings of ring_size entries: (lap tag << 18) | record
    uint16_t* rows;    // wavefront-private image-list rows (PIPE / FETCH visits), 64 rows of 64 ids per wavefront
    u64* recs;
    int n_recs;
    int ring_size;     // power of two >= 2 * n_recs
    int idx_base, n_local;  // this shard's slice [idx_base, idx_base + n_local) of the batch
};

__device__ __forceinline__ u64 bcast_u64(u64 v, int src) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(v & 0xffffffffull), src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(v >> 32), src);
    return ((u64)hi << 32) | lo;
}

// Up to maxn entries of queue q: returns the count n; lanes 0..n-1 get one record each.  Wave-uniform call.
__device__ __noinline__ int sk_pop(SkWs ws, int q, int maxn, int& rec_out) {
    const int lane = lane_id();
    u64* headp = ws.ctl + 16 * q;
    u64* tailp = headp + 8;
    int n = 0;
    u64 h = 0;
    if (lane == 0) {
        for (int tries = 0; tries < 4; tries++) {
            h = ldc(headp);
            const u64 t = ldc(tailp);
            const long long avail = (long long)(t - h);
            if (avail <= 0) { n = 0; break; }
            n = avail < (long long)maxn ? (int)avail : maxn;
            if (atomicCAS(headp, h, h + (u64)n) == h) break;
            n = 0;
        }
    }
    n = bcast_i(n, 0);
    h = bcast_u64(h, 0);
    rec_out = -1;
    if (lane < n) {
        const u64 pos = h + (u64)lane;
        const unsigned* e = ws.rings + (size_t)q * ws.ring_size + (pos & (u64)(ws.ring_size - 1));
        const unsigned want = (unsigned)((pos / (u64)ws.ring_size) + 1ull) & kTagMask;
        unsigned v = 0;
        int spins = 0;
        for (;;) {  // the pusher reserved this position before it wrote it
            v = ldc32(e);
            if ((v >> kSlotBits) == want) break;
            if (++spins > (1 << 22)) { atomicExch((int*)(ws.ctl + kCtlAbort), 2); v = 0; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        rec_out = (int)(v & kSlotMask);
    }
    return n;
}

// Lanes with `pred` append their record to queue q.  Wave-uniform call; the caller drained its record stores.
__device__ __noinline__ void sk_push(SkWs ws, int q, bool pred, int rec) {
    const u64 mask = __ballot(pred);
    if (!mask) return;
    const int lane = lane_id();
    const int n = __popcll(mask);
    const int leader = __ffsll((long long)mask) - 1;
    u64* tailp = ws.ctl + 16 * q + 8;
    u64 base = 0;
    if (lane == leader) base = atomicAdd(tailp, (u64)n);
    base = bcast_u64(base, leader);
    if (pred) {
        const u64 pos = base + (u64)__popcll(mask & ((1ull << lane) - 1ull));
        unsigned* e = ws.rings + (size_t)q * ws.ring_size + (pos & (u64)(ws.ring_size - 1));
        const unsigned tag = (unsigned)((pos / (u64)ws.ring_size) + 1ull) & kTagMask;
        stc32(e, (tag << kSlotBits) | (unsigned)rec);
    }
}

// ---- record <-> registers / LDS ------------------------------------------------------------------------------
// Records are accessed through a raw buffer resource with sc1 (write-through / L1-bypassing) 16-byte accesses
// (__builtin_amdgcn_raw_buffer_load/store_b128, aux = 16: the compiler tracks them like any other load, so a whole
// set of pages is in flight behind one wait).  State pages move COOPERATIVELY: 16 consecutive lanes carry the 16
// chunks of one 256-byte page of one record, so every store instruction writes four whole pages (full cache lines;
// per-lane 16-byte sc1 stores measured 2.7x write amplification at the memory side) and the LDS side is a
// transposition ([index][slot] <-> record order).
typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr int kSc1 = 16;  // aux / cache-policy bit of the buffer intrinsics: sc1
struct RecBuf { __amdgpu_buffer_rsrc_t r; };
__device__ __forceinline__ RecBuf make_recbuf(const SkWs& ws) {
    RecBuf b;
    b.r = __builtin_amdgcn_make_buffer_rsrc((void*)ws.recs, 0, ws.n_recs * (int)(kRecWords * sizeof(u64)), 0x00020000);
    return b;
}
__device__ __forceinline__ v4u rec_load16(RecBuf b, int rec, int word) {
    return __builtin_amdgcn_raw_buffer_load_b128(b.r, rec * (int)(kRecWords * sizeof(u64)) + word * 8, 0, kSc1);
}
__device__ __forceinline__ void rec_store16(RecBuf b, int rec, int word, v4u v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, b.r, rec * (int)(kRecWords * sizeof(u64)) + word * 8, 0, kSc1);
}

__device__ __forceinline__ void load_ps(RecBuf b, int rec, PipeState& ps) {
    v4u q[12];
#pragma unroll
    for (int k = 0; k < 12; k++) q[k] = rec_load16(b, rec, kRecPS + 2 * k);
    __builtin_memcpy(&ps, q, sizeof(ps));
}
__device__ __forceinline__ void store_ps(RecBuf b, int rec, const PipeState& ps) {
    v4u q[12];
    __builtin_memcpy(q, &ps, sizeof(ps));
#pragma unroll
    for (int k = 0; k < 12; k++) rec_store16(b, rec, kRecPS + 2 * k, q[k]);
}

// state word w (record order, run-time) of LDS slot `slot`
__device__ __forceinline__ u64 lds_state_word(int w, int slot) {
    const int di = d_of_rec_word(w);
    if (di >= 0) return (u64)__double_as_longlong(bq_lds_d[di * BQ_SLOTS + slot]);
    if (w >= 70 && w < 77) {
        const int i0 = 2 * (w - 70);
        const unsigned lo = (unsigned)bq_lds_i[i0 * BQ_SLOTS + slot];
        const unsigned hi = (i0 + 1 < BQ_NI) ? (unsigned)bq_lds_i[(i0 + 1) * BQ_SLOTS + slot] : 0u;
        return ((u64)hi << 32) | lo;
    }
    return 0ull;
}
__device__ __forceinline__ void set_lds_state_word(int w, int slot, u64 v) {
    const int di = d_of_rec_word(w);
    if (di >= 0) bq_lds_d[di * BQ_SLOTS + slot] = __longlong_as_double((long long)v);
    else if (w >= 70 && w < 77) {
        const int i0 = 2 * (w - 70);
        bq_lds_i[i0 * BQ_SLOTS + slot] = (int)(unsigned)v;
        if (i0 + 1 < BQ_NI) bq_lds_i[(i0 + 1) * BQ_SLOTS + slot] = (int)(v >> 32);
    }
}
constexpr int kCoopSteps = (SK_P + 3) / 4;  // four records per instruction

// pages 0..npages-1 of the records of slots 0..n-1 (lane s holds the record of slot s) -> LDS.  Wave-uniform call.
__device__ __noinline__ void page_in_coop(RecBuf b, int myrec, int n, int npages) {
    const int lane = lane_id();
    const int sub = lane >> 4, chunk = lane & 15;
    v4u q[kPagesAll][kCoopSteps];
#pragma unroll
    for (int pg = 0; pg < kPagesAll; pg++)
#pragma unroll
        for (int st = 0; st < kCoopSteps; st++) {
            const int slot = 4 * st + sub;
            const int rec = shfl_i(myrec, slot & 63);
            q[pg][st] = v4u{0u, 0u, 0u, 0u};
            if (pg < npages && slot < n) q[pg][st] = rec_load16(b, rec, pg * kPageWords + 2 * chunk);
        }
#pragma unroll
    for (int pg = 0; pg < kPagesAll; pg++)
#pragma unroll
        for (int st = 0; st < kCoopSteps; st++) {
            const int slot = 4 * st + sub;
            if (pg < npages && slot < n) {
                const int w = pg * kPageWords + 2 * chunk;
                set_lds_state_word(w, slot, ((u64)q[pg][st][1] << 32) | q[pg][st][0]);
                set_lds_state_word(w + 1, slot, ((u64)q[pg][st][3] << 32) | q[pg][st][2]);
            }
        }
}
// the pages in `pages` of the slots in `slots` (bit s <=> slot s) LDS -> records.  Wave-uniform call.
__device__ __noinline__ void page_out_coop(RecBuf b, int myrec, u64 slots, unsigned pages) {
    const int lane = lane_id();
    const int sub = lane >> 4, chunk = lane & 15;
#pragma unroll
    for (int pg = 0; pg < kPagesAll; pg++) {
        if (!((pages >> pg) & 1u)) continue;
#pragma unroll
        for (int st = 0; st < kCoopSteps; st++) {
            const int slot = 4 * st + sub;
            const int rec = shfl_i(myrec, slot & 63);
            if (slot < SK_P && ((slots >> slot) & 1ull)) {
                const int w = pg * kPageWords + 2 * chunk;
                const u64 w0 = lds_state_word(w, slot), w1 = lds_state_word(w + 1, slot);
                rec_store16(b, rec, w, v4u{(unsigned)w0, (unsigned)(w0 >> 32), (unsigned)w1, (unsigned)(w1 >> 32)});
            }
        }
    }
}

// image list: record <-> the lane's private row
__device__ __forceinline__ void row_in(const u64* rec, uint16_t* row) {
    u64 w[kRowWords];
#pragma unroll
    for (int k = 0; k < kRowWords; k++) w[k] = ldc(rec + kRecRow + k);
#pragma unroll
    for (int k = 0; k < kRowWords; k++) ((u64*)row)[k] = w[k];
}
__device__ __forceinline__ void row_out(u64* rec, const uint16_t* row) {
#pragma unroll
    for (int k = 0; k < kRowWords; k++) stc(rec + kRecRow + k, ((const u64*)row)[k]);
}
// single state words of a record (the PIPE visit reads xfinal / minf / nevals of a finished optimisation)
__device__ __forceinline__ double rec_d(const u64* rec, int state_index) {
    return __longlong_as_double((long long)ldc(rec + rec_word_of_d(state_index)));
}
__device__ __forceinline__ int rec_i(const u64* rec, int k) { return (int)(ldc(rec + rec_word_of_i(k)) >> (32 * (k & 1))); }

// ---- outputs of a finished patch (PatchOptimizer::optimize copy-out, PatchOptimizer.cpp:86-103) ----------------
__device__ __noinline__ void write_outputs(DevScene sc, DevBatch b, const PipeState& ps, const uint16_t* row, bool ok_,
                                           double fminv, double xf0, double xf1, double xf2, int nevals) {
    const int idx = ps.idx;
    b.ok[idx] = ok_ ? 1 : 0;
    if (ok_) {
        for (int k = 0; k < 3; k++) { b.center[4 * idx + k] = ps.c[k]; b.normal[4 * idx + k] = ps.n[k]; }
        b.center[4 * idx + 3] = 1.0f;
        b.normal[4 * idx + 3] = 0.0f;
        b.n_images[idx] = ps.nimg;
        for (int k = 0; k < ps.nimg && k < b.max_images; k++) b.images[(size_t)idx * b.max_images + k] = row[k];
        if (b.color) {
            const F3 rgb = patch_colour_v(sc, row, ps.nimg, HPMVS_F3(ps.c), ps.scale);
            b.color[3 * idx] = rgb.x; b.color[3 * idx + 1] = rgb.y; b.color[3 * idx + 2] = rgb.z;
        }
        if (b.ncc) b.ncc[idx] = 1.4f;  // PatchOptimizer.cpp:95
    }
    if (b.stage) b.stage[idx] = ok_ ? 0 : ps.fail_stage;
    if (b.fmin) b.fmin[idx] = fminv;
    if (b.x) { b.x[3 * idx] = xf0; b.x[3 * idx + 1] = xf1; b.x[3 * idx + 2] = xf2; }
    if (b.result) b.result[idx] = ps.nl_result;
    if (b.nevals) b.nevals[idx] = nevals;
    if (b.ngrabs) b.ngrabs[idx] = ps.ngrabs;
}

__device__ __forceinline__ void count_done(SkWs ws, bool pred) {
    const u64 m = __ballot(pred);
    if (m && lane_id() == __ffsll((long long)m) - 1) atomicAdd((int*)(ws.ctl + kCtlDone), __popcll(m));
}

// ================================================================================================================
// FETCH: new patches enter (optimize() copy-in + addImages, PatchOptimizer.cpp:78-84, 49-50)
__device__ __noinline__ bool visit_fetch(DevScene sc, DevOptions o, DevBatch b, SkWs ws, uint16_t* row) {
    const int lane = lane_id();
    // records: recycled ones first, then never-used ones
    int rec = -1;
    int m = sk_pop(ws, Q_FREE, 64, rec);
    if (m < 64) {
        int f0 = 0, got = 0;
        if (lane == 0) {
            const int seen = (int)ldc32((const unsigned*)(ws.ctl + kCtlNextSlot));
            if (seen < ws.n_recs) {
                const int want = 64 - m;
                f0 = atomicAdd((int*)(ws.ctl + kCtlNextSlot), want);
                got = ws.n_recs - f0;
                got = got < 0 ? 0 : (got > want ? want : got);
            }
        }
        f0 = bcast_i(f0, 0); got = bcast_i(got, 0);
        if (lane >= m && lane < m + got) rec = f0 + (lane - m);
        m += got;
    }
    if (m == 0) return false;
    // input patches for them
    int i0 = 0;
    if (lane == 0) i0 = atomicAdd((int*)(ws.ctl + kCtlNextIdx), m);
    i0 = bcast_i(i0, 0);
    int valid = ws.n_local - i0;
    valid = valid < 0 ? 0 : (valid > m ? m : valid);
    sk_push(ws, Q_FREE, lane >= valid && lane < m, rec);  // surplus records go back
    if (valid == 0) return false;
    const bool active = lane < valid;
    bool to_eval = false, finished = false;
    PipeState ps;
    __builtin_memset(&ps, 0, sizeof(ps));
    u64* R = ws.recs + (size_t)(active ? rec : 0) * kRecWords;
    const RecBuf rb = make_recbuf(ws);
    if (active) {
        const int idx = ws.idx_base + i0 + lane;
        ps.idx = idx;
        for (int k = 0; k < 3; k++) { ps.c0[k] = b.center[4 * idx + k]; ps.n0[k] = b.normal[4 * idx + k]; ps.c[k] = ps.c0[k]; ps.n[k] = ps.n0[k]; }
        ps.scale = b.scale[idx];
        int nimg = b.n_images[idx];
        ps.refindex = -1;
        do {
            if (nimg < 0) { ps.fail_stage = -nimg; nimg = 0; finished = true; break; }  // rejected by seed_init_kernel
            if (nimg > kMaxImages || nimg > b.max_images) { ps.fail_stage = 100; finished = true; break; }
            bool bad_id = false;
            for (int k = 0; k < nimg; k++) {
                const int id = b.images[(size_t)idx * b.max_images + k];
                bad_id |= (id < 0 || id >= sc.n_views);
                row[k] = (uint16_t)id;
            }
            if (bad_id) { ps.fail_stage = 101; finished = true; break; }
            nimg = add_images_v(sc, o, row, nimg, HPMVS_F3(ps.c), HPMVS_F3(ps.n), ps.scale);
            if (nimg < 0) { ps.fail_stage = 100; finished = true; break; }
            if (nimg < o.MIN_IMAGES_PER_PATCH) { ps.fail_stage = 1; finished = true; break; }
            ps.op = OP_INCC; ps.req_ref = 0; ps.req_robust = 0; ps.req_thr = o.NCC_ALPHA_1;
            ps.stage = T_F1;
            to_eval = true;
        } while (0);
        ps.nimg = nimg;
        if (finished) write_outputs(sc, b, ps, row, false, 0.0, 0.0, 0.0, 0.0, 0);
        else { store_ps(rb, rec, ps); row_out(R, row); }
    }
    drain_stores();
    sk_push(ws, Q_EVAL, to_eval, rec);
    sk_push(ws, Q_FREE, finished, rec);
    count_done(ws, finished);
    return true;
}

// ================================================================================================================
// EVAL: objective_fn / setINCCs requests of up to 64 patches (PatchOptimizer.cpp:286-311, 448-474)
__device__ __noinline__ bool visit_eval(DevScene sc, DevOptions o, SkWs ws) {
    const int lane = lane_id();
    int rec = -1;
    SKV_T(ta);
    const int n = sk_pop(ws, Q_EVAL, 64, rec);
    if (n == 0) return false;
    SKV_T(tb);
    const bool active = lane < n;
    u64* R = ws.recs + (size_t)(active ? rec : 0) * kRecWords;
    PipeState ps;
    __builtin_memset(&ps, 0, sizeof(ps));
    const RecBuf rb = make_recbuf(ws);
    if (active) load_ps(rb, rec, ps);
    const uint16_t* recrow = (const uint16_t*)(R + kRecRow);
    PackedReq q;
    q.obj_mode = (ps.op == OP_OBJ) ? 1 : 0;
    q.ref = ps.req_ref; q.robust = ps.req_robust; q.nimg = ps.nimg; q.thr = ps.req_thr; q.scale = ps.scale;
    {   // calculatePatchAxis for the request's reference image (PatchOptimizer.cpp:532-548), one lane per request
        float ax[3] = {0, 0, 0}, ay[3] = {0, 0, 0}, az[3] = {0, 0, 1};
        if (active) {
            const DevView& RV = sc.views[list_id<true>(recrow, ps.op == OP_OBJ ? 0 : ps.req_ref)];
            calc_patch_axes(RV.xaxis, RV.yaxis, ps.n, ps.scale, ax, ay, az);
        }
        for (int k = 0; k < 3; k++) { q.c[k] = ps.c[k]; q.zax[k] = (ps.op == OP_OBJ) ? az[k] : ps.n[k]; q.px[k] = ax[k]; q.py[k] = ay[k]; }
    }
    PackedRes r;
    r.obj = 2.0; r.keep = 0ull; r.incc_sum = 0.0f; r.ngrabs = 0;
    const u64 pending = __ballot(active && ps.op != OP_NONE);
    SKV_T(tc);
#ifndef SK_EVAL_COMPACT
    // 49 lanes per grab, sums through LDS: neighbouring lanes read neighbouring texels, 2.1 k wave-cycles per grab
    wave_eval_packed<true>(sc, o, recrow, pending, q, r, *reinterpret_cast<EvalLds*>(sk_lds));
#else
    // one lane per usable (request, image) grab: 4x fewer instructions but every lane gathers from its own cache
    // lines (13x L1 fill amplification): 3.4 k wave-cycles per grab measured (profiles/r02/stream_kernel)
    r = wave_eval_compact<true>(sc, o, recrow, pending, q, r);
#endif
    SKV_T(td);
    if (active) {
        ps.res_f = r.obj; ps.res_mask = r.keep; ps.res_sum = r.incc_sum; ps.ngrabs += r.ngrabs; ps.op = OP_NONE;
        // only the result words change: ngrabs (word 2), op (word 4), res_sum (word 7), res_f (22), res_mask (23)
        u64 w[24];
        __builtin_memcpy(w, &ps, sizeof(ps));
        stc(R + kRecPS + 2, w[2]); stc(R + kRecPS + 4, w[4]); stc(R + kRecPS + 7, w[7]);
        stc(R + kRecPS + 22, w[22]); stc(R + kRecPS + 23, w[23]);
    }
    drain_stores();
    sk_push(ws, Q_AE, active && ps.stage == T_BQ, rec);
    sk_push(ws, Q_PIPE, active && ps.stage != T_BQ, rec);
    SKV_T(te);
    SKV_ACC(ws, 50, ta, tb); SKV_ACC(ws, 51, tb, tc); SKV_ACC(ws, 52, tc, td); SKV_ACC(ws, 53, td, te);
    SKV_ACC(ws, 54, 0, (long long)n); SKV_ACC(ws, 55, 0, 1ll);
    return true;
}

// ================================================================================================================
// block visits of the optimiser: AE (code after an evaluation), L60 (trsbox), L90 (xbase shift), L210 (altmov +
// L230), L230, RESC (rescue_)
__device__ __noinline__ bool visit_block(DevScene sc, DevOptions o, SkWs ws, int q) {
    const int lane = lane_id();
    int rec = -1;
    SKV_T(ta);
    const int n = sk_pop(ws, q, SK_P, rec);
    if (n == 0) return false;
    SKV_T(tb);
    const bool active = lane < n;
    u64* R = ws.recs + (size_t)(active ? rec : 0) * kRecWords;
    Bobyqa3 B;
    B.slot = lane < SK_P ? lane : 0;
    PipeState ps;
    __builtin_memset(&ps, 0, sizeof(ps));
    // what this block reads (a prefix of the record) and may write (pages): see the record layout
    const int in_pages = (q == Q_L60) ? 4 : kPagesAll;
    const unsigned out_pages = (q == Q_L60) ? 0x07u : (q == Q_L90) ? 0x1fu : (q == Q_L210 || q == Q_L230) ? 0x27u : 0x3fu;
#ifdef SK_POISON
    for (int k = lane; k < kSkLdsBytes / 8; k += 64) ((u64*)sk_lds)[k] = 0xfff8dead0000beefull;  // NaN: a page-set bug shows
    wave_lds_sync();
#endif
    const RecBuf rb = make_recbuf(ws);
    page_in_coop(rb, rec, n, in_pages);
    if (active) load_ps(rb, rec, ps);
    wave_lds_sync();
    // (AE rewrites zmat, whose tail shares page 5 with vlag / wsuma: all six pages go back)
    SKV_T(tc);
    int label = bq::LRET_DONE;
    constexpr unsigned cheap = (1u << bq::L20) | (1u << bq::L650) | (1u << bq::L680) | (1u << bq::L360) | (1u << bq::L720);
    unsigned allowed = cheap;
    if (active) {
        switch (q) {
        case Q_AE: label = bq_step_entry(B, ps.res_f); break;
        case Q_L60: label = bq::step_label(B, bq::L60); break;
        case Q_L90: label = bq::step_label(B, bq::L90); break;
        case Q_L210: label = bq::step_label(B, bq::L210); break;
        case Q_L230: label = bq::L230; break;
        default: label = bq::step_label(B, bq::L190); break;
        }
    }
    if (q == Q_L210 || q == Q_L230) allowed |= 1u << bq::L230;  // altmov is always followed by L230; L230 may repeat once (xalt)
    for (;;) {
        label = bq::run_set(B, label, allowed);
        const bool skip90 = active && label == bq::L90 && bq::l90_is_noop(B);
        if (skip90) label = bq::l90_next(B);
        if (!__any(skip90)) break;
    }
    // ---- where to next
    SKV_T(td);
    int nq = -1;
    if (active) {
        if (label == bq::LRET_CONT) {  // next objective evaluation: setCenterNorm(x) (PatchOptimizer.cpp:401-414)
            OptFields F;
            __builtin_memcpy(&F, ps.F, sizeof(F));
            const double xe[3] = {BQA(xeval, 0), BQA(xeval, 1), BQA(xeval, 2)};
            set_center_norm(F, o, xe, ps.c, ps.n);
            ps.op = OP_OBJ;
            nq = Q_EVAL;
        } else if (label == bq::LRET_DONE) {
            ps.stage = T_BQDONE;
            ps.nl_result = BQN(rc);
            nq = Q_PIPE;
        } else if (label == bq::L60) nq = Q_L60;
        else if (label == bq::L90) nq = Q_L90;
        else if (label == bq::L210) nq = Q_L210;
        else if (label == bq::L230) nq = Q_L230;
        else if (label == bq::L190) nq = Q_RESC;
        else atomicExch((int*)(ws.ctl + kCtlAbort), 3);
        store_ps(rb, rec, ps);
    }
    wave_lds_sync();
    page_out_coop(rb, rec, n >= 64 ? ~0ull : ((1ull << n) - 1ull), out_pages);
    drain_stores();
    SKV_T(te);
    sk_push(ws, Q_EVAL, nq == Q_EVAL, rec);
    sk_push(ws, Q_L60, nq == Q_L60, rec);
    sk_push(ws, Q_L230, nq == Q_L230, rec);
    sk_push(ws, Q_L210, nq == Q_L210, rec);
    sk_push(ws, Q_L90, nq == Q_L90, rec);
    sk_push(ws, Q_PIPE, nq == Q_PIPE, rec);
    sk_push(ws, Q_RESC, nq == Q_RESC, rec);
    wave_lds_sync();
    SKV_T(tf);
    SKV_ACC(ws, 40, ta, tb); SKV_ACC(ws, 41, tb, tc); SKV_ACC(ws, 42, tc, td); SKV_ACC(ws, 43, td, te); SKV_ACC(ws, 44, te, tf);
    if (lane_id() == 0) { SKV_ACC(ws, 45, 0, (long long)n); SKV_ACC(ws, 46, 0, 1ll); }
    return true;
}

// ================================================================================================================
// PIPE: the image-list passes of runOptimization between the photometric requests (PatchOptimizer.cpp:48-76)
__device__ __noinline__ bool visit_pipe(DevScene sc, DevOptions o, DevBatch b, SkWs ws, uint16_t* row) {
    const int lane = lane_id();
    int rec = -1;
    const int n = sk_pop(ws, Q_PIPE, SK_P, rec);
    if (n == 0) return false;
    const bool active = lane < n;
    u64* R = ws.recs + (size_t)(active ? rec : 0) * kRecWords;
    Bobyqa3 B;
    B.slot = lane < SK_P ? lane : 0;
    PipeState ps;
    __builtin_memset(&ps, 0, sizeof(ps));
    bool to_eval = false, fail = false, okdone = false, started = false;
    const double lbv[3] = {-HUGE_VAL, -23.99999, -23.99999};  // PatchOptimizer.cpp:326-336
    const double ubv[3] = {HUGE_VAL, 23.99999, 23.99999};
    const RecBuf rb = make_recbuf(ws);
    if (active) {
        load_ps(rb, rec, ps);
        row_in(R, row);
        int nimg = ps.nimg;
        switch (ps.stage) {
        case T_F1: {  // filterImagesNCC(alpha1) result -> sortImages, assureImageAngles, optimizePatch start
            nimg = apply_keep_mask(row, nimg, ps.res_mask);
            if (nimg < o.MIN_IMAGES_PER_PATCH) { ps.fail_stage = 2; fail = true; break; }
            nimg = sort_images_v(sc, o, row, nimg, HPMVS_F3(ps.c), HPMVS_F3(ps.n), ps.scale);
            if (!assure_image_angles_v(sc, o, row, nimg, HPMVS_F3(ps.c))) { ps.fail_stage = 3; fail = true; break; }
            if (nimg < o.MIN_IMAGES_PER_PATCH) { ps.fail_stage = 4; fail = true; break; }
            OptFields F;
            set_optimization_fields(sc.views[row[0]], ps.c, F);
            __builtin_memcpy(ps.F, &F, sizeof(F));
            double x0[3], dx[3];
            parameters_from_center_norm(F, o, ps.n, lbv, ubv, x0);
            bq_default_step(x0, lbv, ubv, dx);
            started = true;
            ps.have_bq = 1;
            if (bq_start(B, x0, lbv, ubv, dx, 1.e-7, 1000) != BQ_CONTINUE) {
                ps.nl_result = BQN(rc); ps.fail_stage = 4; fail = true; break;
            }
            { const double xe[3] = {BQA(xeval, 0), BQA(xeval, 1), BQA(xeval, 2)}; set_center_norm(F, o, xe, ps.c, ps.n); }
            ps.op = OP_OBJ;
            ps.stage = T_BQ;
            to_eval = true;
            break;
        }
        case T_BQDONE: {  // optimizePatch result (PatchOptimizer.cpp:367-381) -> addImages, filterImagesNCC(alpha2)
            // SUCCESS / STOPVAL / FTOL / XTOL only; negative codes throw in the reference
            if (!(ps.nl_result >= 1 && ps.nl_result <= 4)) { ps.fail_stage = 4; fail = true; break; }
            {
                OptFields F;
                __builtin_memcpy(&F, ps.F, sizeof(F));
                const double xf[3] = {rec_d(R, O_xfinal), rec_d(R, O_xfinal + 1), rec_d(R, O_xfinal + 2)};
                set_center_norm(F, o, xf, ps.c, ps.n);
            }
            nimg = add_images_v(sc, o, row, nimg, HPMVS_F3(ps.c), HPMVS_F3(ps.n), ps.scale);
            if (nimg < 0) { ps.fail_stage = 100; fail = true; break; }
            if (nimg < o.MIN_IMAGES_PER_PATCH) { ps.fail_stage = 5; fail = true; break; }
            ps.op = OP_INCC; ps.req_ref = 0; ps.req_robust = 0; ps.req_thr = o.NCC_ALPHA_2;
            ps.stage = T_F2;
            to_eval = true;
            break;
        }
        case T_F2: {  // filterImagesNCC(alpha2) -> filterImagesByAngle, assureImageAngles, setRefImage start
            nimg = apply_keep_mask(row, nimg, ps.res_mask);
            if (nimg < o.MIN_IMAGES_PER_PATCH) { ps.fail_stage = 6; fail = true; break; }
            nimg = filter_images_by_angle_v(sc, o, row, nimg, HPMVS_F3(ps.c), HPMVS_F3(ps.n));
            if (nimg < o.MIN_IMAGES_PER_PATCH) { ps.fail_stage = 7; fail = true; break; }
            if (!assure_image_angles_v(sc, o, row, nimg, HPMVS_F3(ps.c))) { ps.fail_stage = 8; fail = true; break; }
            ps.ii = 0; ps.refindex = -1; ps.refbest = 3.402823466e+38f;
            ps.op = OP_REFSUM; ps.req_ref = 0; ps.req_robust = 1;
            ps.stage = T_RS;
            to_eval = true;
            break;
        }
        case T_RS: {  // setRefImage (PatchOptimizer.cpp:154-181): one robust setINCCs per candidate
            if (ps.res_sum < ps.refbest) { ps.refbest = ps.res_sum; ps.refindex = ps.ii; }
            ps.ii++;
            if (ps.ii < nimg) { ps.op = OP_REFSUM; ps.req_ref = ps.ii; ps.req_robust = 1; to_eval = true; break; }
            if (ps.refindex > 0) {  // swap the chosen image into slot 0
                const uint16_t t = row[0];
                row[0] = row[ps.refindex];
                row[ps.refindex] = t;
            }
            ps.op = OP_INCC; ps.req_ref = 0; ps.req_robust = 0; ps.req_thr = o.NCC_ALPHA_2;
            ps.stage = T_F3;
            to_eval = true;
            break;
        }
        case T_F3: {
            nimg = apply_keep_mask(row, nimg, ps.res_mask);
            if (nimg < o.MIN_IMAGES_PER_PATCH) { ps.fail_stage = 9; fail = true; break; }
            if (nimg > b.max_images) { ps.fail_stage = 100; fail = true; break; }
            okdone = true;
            break;
        }
        default: atomicExch((int*)(ws.ctl + kCtlAbort), 4); fail = true; break;
        }
        ps.nimg = nimg;
        if (fail || okdone) {
            double fminv = 0.0, xf0 = 0.0, xf1 = 0.0, xf2 = 0.0;
            int nev = 0;
            if (started) { fminv = BQS(minf); xf0 = BQA(xfinal, 0); xf1 = BQA(xfinal, 1); xf2 = BQA(xfinal, 2); nev = BQN(nevals); }
            else if (ps.have_bq) {
                fminv = rec_d(R, O_minf);
                xf0 = rec_d(R, O_xfinal); xf1 = rec_d(R, O_xfinal + 1); xf2 = rec_d(R, O_xfinal + 2);
                nev = rec_i(R, I_nevals);
            }
            write_outputs(sc, b, ps, row, okdone, fminv, xf0, xf1, xf2, nev);
        } else {
            store_ps(rb, rec, ps);
            row_out(R, row);
        }
    }
    wave_lds_sync();
    page_out_coop(rb, rec, __ballot(active && started && !(fail || okdone)), 0x3fu);  // bq_start wrote the whole state
    drain_stores();
    sk_push(ws, Q_EVAL, to_eval, rec);
    sk_push(ws, Q_FREE, fail || okdone, rec);
    count_done(ws, fail || okdone);
    wave_lds_sync();
    return true;
}

}  // namespace

// ================================================================================================================
#ifdef SK_STATS
#define SK_T0() const long long t0_ = clock64()
#define SK_ADD(i, v) st_[(i)] += (v)
#define SK_TADD(i) st_[(i)] += clock64() - t0_
#else
#define SK_T0() do { } while (0)
#define SK_ADD(i, v) do { } while (0)
#define SK_TADD(i) do { } while (0)
#endif
// The batch is cut into kShards independent slices, each with its own queues, records and wavefronts (shard =
// blockIdx % kShards, which on this part is also the XCD a workgroup lands on -- a locality bonus, nothing the
// correctness depends on): 8x fewer wavefronts contend for a queue head, and a slice's records stay in one L2.
constexpr int kShards = 8;
struct SkLaunch {
    u64* ctl; unsigned* rings; uint16_t* rows; u64* recs;
    int n_recs_shard, ring_size;
    size_t ctl_stride_words, ring_stride, rec_stride_words;
};

__global__ void __launch_bounds__(64, 2) stream_kernel(DevScene sc, DevOptions o, DevBatch b, SkLaunch L) {
    const int lane = lane_id();
    const int shard = (int)(blockIdx.x % kShards);
    SkWs ws;
    ws.ctl = L.ctl + (size_t)shard * L.ctl_stride_words;
    ws.rings = L.rings + (size_t)shard * L.ring_stride;
    ws.rows = L.rows;
    ws.recs = L.recs + (size_t)shard * L.rec_stride_words;
    ws.n_recs = L.n_recs_shard;
    ws.ring_size = L.ring_size;
    {
        const int base = b.n / kShards, rem = b.n % kShards;
        ws.idx_base = shard * base + (shard < rem ? shard : rem);
        ws.n_local = base + (shard < rem ? 1 : 0);
    }
    uint16_t* row = ws.rows + ((size_t)blockIdx.x * 64 + lane) * 64;
    int idle = 0;
    unsigned rng = (unsigned)blockIdx.x * 2654435761u + 12345u;
#ifdef SK_STATS
    long long st_[40];
    for (int k = 0; k < 40; k++) st_[k] = 0;
    const long long tstart_ = clock64();
    if (blockIdx.x == 0 && lane == 0) g_evdbg = L.ctl + kCtlStats + 56;  // evaluator timers land in shard 0's block
    __syncthreads();
#endif
    for (;;) {
        // ---- queue fill levels, one lane per queue
        long long avail = 0;
        if (lane < Q_COUNT) {
            const u64 h = ldc(ws.ctl + 16 * lane), t = ldc(ws.ctl + 16 * lane + 8);
            avail = (long long)(t - h);
            if (avail < 0) avail = 0;
        }
        int misc = 0;  // lane 0: next input, lane 1: next record, lane 2: done, lane 3: abort
        if (lane < 4) misc = (int)ldc32((const unsigned*)(ws.ctl + kCtlNextIdx + 8 * lane));
        const int next_idx = bcast_i(misc, 0), next_rec = bcast_i(misc, 1), done = bcast_i(misc, 2), abort_ = bcast_i(misc, 3);
        if (abort_ != 0 || done >= ws.n_local) break;
        const int av = (int)(avail > 0x7fffffffll ? 0x7fffffffll : avail);
        const int a_free = bcast_i(av, Q_FREE);
        const bool input_left = next_idx < ws.n_local;
        const bool fetch_ok = input_left && (a_free >= 32 || next_rec < ws.n_recs);
        bool did = false;
        SK_T0();
        if (fetch_ok) { did = visit_fetch(sc, o, b, ws, row); if (did) { SK_ADD(Q_FREE, 1); SK_TADD(16 + Q_FREE); } }
        if (!did) {
            // queues that can fill a whole visit (64 requests for EVAL, SK_P patches otherwise): take one of them at
            // random, so that the wavefronts of a shard do not all rush to the same queue head; otherwise the fullest
            const int cap = (lane == Q_EVAL) ? 64 : SK_P;
            const u64 ready = __ballot(lane < Q_FREE && av >= cap);
            int bestq = -1, best = 0;
            if (ready) {
                rng = rng * 1664525u + 1013904223u;
                int k = (int)((rng >> 16) % (unsigned)__popcll(ready));
                u64 m = ready;
                while (k-- > 0) m &= m - 1ull;
                bestq = __ffsll((long long)m) - 1;
                best = 1;
            } else {
                int score = 0;
                if (lane < Q_FREE) score = (lane == Q_EVAL) ? (av * SK_P) : (av * 64);
                best = score; bestq = lane;
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    const int os = __shfl_xor(best, off), oq = __shfl_xor(bestq, off);
                    if (os > best || (os == best && oq < bestq)) { best = os; bestq = oq; }
                }
                best = bcast_i(best, 0); bestq = bcast_i(bestq, 0);
            }
            if (best > 0) {
                if (bestq == Q_EVAL) did = visit_eval(sc, o, ws);
                else if (bestq == Q_PIPE) did = visit_pipe(sc, o, b, ws, row);
                else did = visit_block(sc, o, ws, bestq);
                if (did) { SK_ADD(bestq, 1); SK_TADD(16 + bestq); } else { SK_ADD(10, 1); SK_TADD(26); }
            } else if (input_left && a_free > 0) {
                did = visit_fetch(sc, o, b, ws, row);
            }
        }
        if (did) idle = 0;
        else {
            SK_ADD(11, 1);
            if (++idle > (1 << 22)) { if (lane == 0) atomicExch((int*)(ws.ctl + kCtlAbort), 1); break; }
            // back off: the fewer hits, the longer the nap (up to ~127 * 64 clocks)
            if (idle < 4) __builtin_amdgcn_s_sleep(4);
            else if (idle < 16) __builtin_amdgcn_s_sleep(16);
            else if (idle < 64) __builtin_amdgcn_s_sleep(48);
            else { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); }
            SK_TADD(27);
        }
    }
#ifdef SK_STATS
    st_[28] = clock64() - tstart_;
    if (lane == 0)
        for (int k = 0; k < 40; k++) atomicAdd(ws.ctl + kCtlStats + k, (u64)st_[k]);
#endif
}

// ---- host side -------------------------------------------------------------------------------------------------
static int sk_waves(int n_cus) {
    int per_cu = (160 * 1024) / (kSkLdsBytes + 64);
    if (per_cu > 8) per_cu = 8;
    return n_cus * per_cu;
}
static int sk_ring_size(int n_recs) {
    int r = 64;
    while (r < 2 * n_recs) r <<= 1;
    return r;
}
static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static size_t sk_ctl_stride_words() { return align_up(sizeof(u64) * kCtlWords, 4096) / sizeof(u64); }

// in-flight patch records of ONE shard for a batch of n patches
int stream_records_for(int n) {
    int cap = 131072;  // all shards together: ~5x the lanes of all resident wavefronts, 235 MB of records
    if (const char* e = getenv("HPMVS_STREAM_SLOTS")) { const int v = atoi(e); if (v >= 64 * kShards) cap = v; }
    cap /= kShards;
    if (cap > (1 << kSlotBits)) cap = 1 << kSlotBits;
    const int per_shard = (n + kShards - 1) / kShards;
    return per_shard < cap ? (per_shard < 64 ? 64 : per_shard) : cap;
}
size_t stream_workspace_zero_bytes(int n_recs) {
    return kShards * (sk_ctl_stride_words() * sizeof(u64) + (size_t)Q_COUNT * sk_ring_size(n_recs) * sizeof(unsigned));
}
size_t stream_workspace_bytes(int n_cus, int n_recs) {
    size_t sz = align_up(stream_workspace_zero_bytes(n_recs), 4096);
    sz += align_up((size_t)sk_waves(n_cus) * 64 * 64 * sizeof(uint16_t), 4096);
    sz += align_up((size_t)kShards * n_recs * kRecWords * sizeof(u64), 4096);
    return sz;
}
size_t stream_abort_offset() { return sizeof(u64) * kCtlAbort; }
size_t stream_stats_offset() { return sizeof(u64) * kCtlStats; }
size_t stream_shard_stride() { return sk_ctl_stride_words() * sizeof(u64); }
int stream_shards() { return kShards; }

void launch_stream(const DevScene& sc, const DevOptions& o, const DevBatch& b, void* workspace, int n_cus, int n_recs,
                   hipStream_t st) {
    if (b.n <= 0) return;
    SkLaunch L;
    char* p = (char*)workspace;
    L.ctl = (u64*)p;
    L.ctl_stride_words = sk_ctl_stride_words();
    p += kShards * L.ctl_stride_words * sizeof(u64);
    L.ring_size = sk_ring_size(n_recs);
    L.ring_stride = (size_t)Q_COUNT * L.ring_size;
    L.rings = (unsigned*)p;
    p = (char*)workspace + align_up(stream_workspace_zero_bytes(n_recs), 4096);
    L.rows = (uint16_t*)p; p += align_up((size_t)sk_waves(n_cus) * 64 * 64 * sizeof(uint16_t), 4096);
    L.recs = (u64*)p;
    L.rec_stride_words = (size_t)n_recs * kRecWords;
    L.n_recs_shard = n_recs;
    int waves = sk_waves(n_cus);
    int need = ((b.n + SK_P - 1) / SK_P) * 3;  // small batches: no more wavefronts than could ever be busy
    need = (need + kShards - 1) / kShards * kShards;
    if (waves > need) waves = need < kShards ? kShards : need;
    hipLaunchKernelGGL(stream_kernel, dim3(waves), dim3(64), 0, st, sc, o, b, L);
}

}  // namespace hpmvs
